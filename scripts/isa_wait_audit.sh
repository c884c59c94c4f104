# Lists, per kernel of one .hip file, the global loads that are waited for immediately (s_waitcnt vmcnt(0) within three
# instructions of a non-LDS global_load): each is a full memory round trip on the critical path of the wave.  DESIGN_LOG.md 10.8.
#   usage: bash scripts/isa_wait_audit.sh maua_amd/csrc/modconv_dma.hip
set -e
src=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -Iinclude --cuda-device-only -S "$src" -o /tmp/_audit.s 2>/dev/null
awk '/^_Z[A-Za-z0-9_]*:/ {name=$1}
     /global_load_(dword|ushort|ubyte|short)/ && !/lds/ {loads[name]++; pend=3; next}
     pend > 0 { if ($0 ~ /s_waitcnt vmcnt\(0\)/) {hot[name]++; pend=0} else pend-- }
     END {for (k in loads) printf "%5d loads %5d waited-at-once  %s\n", loads[k], hot[k], substr(k, 1, 110)}' /tmp/_audit.s | sort -k3 -n -r

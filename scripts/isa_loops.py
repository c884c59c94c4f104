"""Instruction mix of the MFMA-carrying basic blocks of one kernel (gfx950): what a step / stage loop really issues.
python scripts/isa_loops.py file.hip kernel-name-substring [min_mfma] [extra hipcc flags ...]
Prints one line per basic block with >= min_mfma MFMAs: counts of MFMA, VALU, LDS, VMEM, SALU, lane spills (v_readlane /
v_writelane), scratch accesses, waitcnts, barriers."""
import collections, os, re, subprocess, sys, tempfile

src, pat = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
extra = sys.argv[4:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    f"-I{root}/include", f"-I{root}/maua_amd/csrc", *extra, src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
KEYS = (("v_mfma", "mfma"), ("v_readlane", "lanespill"), ("v_writelane", "lanespill"), ("v_pk_", "vpk"), ("v_cvt_pk_bf16", "cvtbf"),
        ("v_", "valu"), ("ds_", "lds"), ("global_", "vmem"), ("buffer_", "vmem"), ("scratch_", "scratch"), ("s_waitcnt", "waitcnt"),
        ("s_barrier", "barrier"), ("s_cbranch", "branch"), ("s_nop", "nop"), ("s_", "salu"))
name, blk, c, on = None, None, collections.Counter(), False
def flush():
    if on and c["mfma"] >= min_mfma:
        print(f"  {blk:12s} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
for line in txt.split("\n"):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        flush()
        name = m.group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        on = pat in dem
        if on:
            print(dem[:110])
        blk, c = "entry", collections.Counter()
        continue
    m = re.match(r"^(\.LBB\w+):", line)
    if m:
        flush()
        blk, c = m.group(1), collections.Counter()
        continue
    t = line.strip()
    if not on or not t or t[0] in ".;" or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_endpgm":
        flush()
        on = False
        continue
    for pfx, key in KEYS:
        if op.startswith(pfx):
            c[key] += 1
            break

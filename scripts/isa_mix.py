"""Static instruction mix per kernel of one HIP source (gfx950): python scripts/isa_mix.py maua_amd/csrc/x.hip
Counts are static (loops counted once) — a quick look at what an epilogue or a staging loop costs in VALU terms."""
import collections, re, subprocess, sys, tempfile, os

src = sys.argv[1]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src,
                    "-o", out], check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
name = None
c = collections.Counter()
for line in txt.split("\n"):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, c = m.group(1), collections.Counter()
        continue
    t = line.strip()
    if not name or not t or t[0] in ".;" or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_endpgm":
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(dem[:70].ljust(70), " ".join(f"{k}={v}" for k, v in sorted(c.items())))
        name = None
        continue
    for pfx, key in (("v_mfma", "mfma"), ("v_cvt_pk_bf16", "cvt_bf16"), ("v_", "valu"), ("ds_", "lds"), ("global_", "vmem"),
                     ("buffer_", "vmem"), ("scratch_", "scratch"), ("s_waitcnt", "waitcnt"), ("s_barrier", "barrier"),
                     ("s_cbranch", "branch"), ("s_", "salu")):
        if op.startswith(pfx):
            c[key] += 1
            break

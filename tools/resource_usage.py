"""Per-kernel register / scratch / occupancy table of the engine (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
  python tools/resource_usage.py [substring ...]      (no argument: kernels with scratch or occupancy < 2, and every MFMA kernel)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "lightkrylov_amd", "csrc", "lk_engine.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                      "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src], capture_output=True, text=True).stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", out)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
want = sys.argv[1:]
print("VGPR AGPR scratch occ LDS  kernel")
for b, d in zip(blocks, dem):
    g = lambda k: int((re.search(k + r": (\d+)", b) or [0, 0])[1])   # noqa: E731
    v, a, s, o, l = g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    d = d.replace("void lk::", "").split("(")[0]
    if (want and any(w in d for w in want)) or (not want and (s > 0 or o < 2 or "mfma" in d)):
        print(f"{v:4d} {a:4d} {s:7d} {o:3d} {l:5d}  {d[:140]}")

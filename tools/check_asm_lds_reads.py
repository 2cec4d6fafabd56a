"""panel_gram_rs reads its MFMA operands with inline-asm ds_read_b64 (one per operand: the compiler would merge pairs into a bank-conflicting ds_read2st64_b64).  An asm load is
invisible to the compiler's wait counting: between the first read of a batch and OUR `s_waitcnt lgkmcnt(0)` nothing may touch a destination register whose data is still on its
way.  This script compiles every instantiation to assembly (hipcc -S, a few seconds) and checks exactly that; it prints one line per kernel and exits non-zero on a violation.
  python tools/check_asm_lds_reads.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INST = [(1, 8, 8), (2, 5, 8), (3, 3, 8), (4, 3, 6), (5, 3, 4), (6, 3, 4), (7, 4, 2), (8, 4, 2)]          # panel_gram_rs as dispatched by lk_engine.hip (dots_mfma)
INST3M = [(1, 8, 8), (2, 5, 6), (3, 3, 4), (4, 5, 2), (5, 5, 2)]                                          # panel_gram_rs3m (complex: ds_read_b128)
INST3M4 = [(6, 5, 2), (7, 5, 2)]                                                    # panel_gram_rs3m4 (four groups: 2 steps x (main loop + ragged tile) x 4 groups = 16 batches)


def main():
    eng = open(os.path.join(ROOT, "lightkrylov_amd", "csrc", "lk_engine.hip")).read()
    for kp, nb, w in INST:
        assert f"panel_gram_rs<{kp}, {nb}, {w}>" in eng, f"instantiation <{kp}, {nb}, {w}> is not the one the engine launches"
    for kp, nb, w in INST3M:
        assert f"panel_gram_rs3m<{kp}, {nb}, {w}>" in eng, f"instantiation 3m <{kp}, {nb}, {w}> is not the one the engine launches"
    for kp, nb, w in INST3M4:
        assert f"panel_gram_rs3m4<{kp}, {nb}, {w}>" in eng, f"instantiation 3m4 <{kp}, {nb}, {w}> is not the one the engine launches"
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.hip")
        with open(src, "w") as f:
            f.write('#include "%s"\n' % os.path.join(ROOT, "lightkrylov_amd", "csrc", "lk_kernels.hip.h"))
            for kp, nb, w in INST:
                f.write(f"template __global__ void lk::panel_gram_rs<{kp}, {nb}, {w}>(const double *, int64_t, int, int64_t, double *);\n")
            for kp, nb, w in INST3M:
                f.write(f"template __global__ void lk::panel_gram_rs3m<{kp}, {nb}, {w}>(const double *, int64_t, int, int64_t, double *);\n")
            for kp, nb, w in INST3M4:
                f.write(f"template __global__ void lk::panel_gram_rs3m4<{kp}, {nb}, {w}>(const double *, int64_t, int, int64_t, double *);\n")
        out = os.path.join(d, "t.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:]); return 2
        lines_all = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines_all) if re.match(r"^_ZN2lk1[356]panel_gram_rs(3m4?)?I\w+:", l)]
    assert len(starts) == len(INST) + len(INST3M) + len(INST3M4), (len(starts), len(INST), len(INST3M), len(INST3M4))
    rc = 0
    for s0 in starts:
        e0 = next(i for i in range(s0, len(lines_all)) if lines_all[i].startswith(".Lfunc_end"))
        lines = lines_all[s0:e0]
        i, regions, bad = 0, 0, []
        while i < len(lines):
            if re.search(r"ds_read_b(64|128)", lines[i]) and "ASMSTART" in lines[i - 1]:
                dests, j, others = set(), i, []
                while j < len(lines) and not ("s_waitcnt lgkmcnt(0)" in lines[j] and "ASMSTART" in lines[j - 1]):
                    l = lines[j].strip()
                    if l.startswith("ds_read_b64") or l.startswith("ds_read_b128"):
                        m = re.match(r"ds_read_b\d+ v\[(\d+):(\d+)\]", l)
                        dests.update(range(int(m.group(1)), int(m.group(2)) + 1))
                    elif l and not l.startswith(";"):
                        others.append((l, set(dests)))
                    j += 1
                assert j < len(lines), "a batch of asm reads without its wait"
                regions += 1
                for o, dset in others:
                    regs = set()
                    for m in re.finditer(r"v\[(\d+):(\d+)\]", o):
                        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
                    for m in re.finditer(r"\bv(\d+)\b", o):
                        regs.add(int(m.group(1)))
                    if regs & dset:
                        bad.append(o)
                i = j
            i += 1
        print(f"{lines[0].split(':')[0]}: {regions} batches of asm reads, {len(bad)} instructions touching a pending destination")
        if regions != (16 if "rs3m4" in lines[0] else 4) or bad:
            rc = 1
            for o in bad[:8]:
                print("   ", o)
    return rc


if __name__ == "__main__":
    sys.exit(main())

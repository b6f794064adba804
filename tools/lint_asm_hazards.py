"""Lint of the hand-placed instruction streams (attn_pipe.hip, gemm_fr.hip): hipcc's hazard recogniser treats an `asm volatile` statement as
opaque, so neither side of a hazard that crosses its boundary gets wait states.  Compiles the file with the product build's flags
(-save-temps) and reports
  A. every asm MFMA whose A / B / C operand overlaps the destination of a compiler-emitted vector instruction fewer than 2 wait states
     ahead (gfx950: VALU write -> MFMA source read needs 2);
  B. every compiler-emitted instruction that reads or writes a register of an asm MFMA's D fewer than 12 (32x32x16, 8 passes) or 8
     (16x16x32, 4 passes) wait states behind it, other than an MFMA taking D whole as its C (XDL write -> VALU / memory read or write:
     passes + 3, one spare).  An instruction counts as one wait state, `s_nop N` as N + 1.
Exit code 1 when any is found.   python tools/lint_asm_hazards.py [file.hip] [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from walkgpt_amd import _build  # noqa: E402  (the flags the product objects are built with)
src = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "walkgpt_amd", "csrc", "attn_pipe.hip")
tmp = tempfile.mkdtemp()
subprocess.run([_build.HIPCC] + _build.FLAGS + ["-save-temps=obj", "-c", src, "-o", os.path.join(tmp, "x.o")] + sys.argv[2:],
               check=True, capture_output=True, cwd=tmp)
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
def mfma_states(op):
    return 12 if "32x32" in op else 8
bad, in_asm, hist, kernel = 0, False, [], "?"
pend = []          # asm MFMA results still inside their window: [registers, wait states left, text]
for ln in open(os.path.join(tmp, asm)):
    t = ln.strip()
    if t.startswith("_Z") and ":" in t and t.split(":")[0].isidentifier():
        kernel = t.split(":")[0][:60]
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    ops = [o.strip() for o in t[len(op):].split(",")]
    if in_asm and op.startswith("v_mfma"):
        srcs = set().union(*[regs(o.split()[0]) for o in ops[1:4] if o])
        states = 0
        for (hop, hdst, hasm, hstates) in reversed(hist):
            if states >= 2:
                break
            if hop.startswith("v_") and not hasm and hdst & srcs:
                print("%s: compiler `%s` writes v%s %d wait state(s) before asm `%s`" % (kernel, hop, sorted(hdst & srcs), states, t[:70]))
                bad += 1
            states += hstates
    w = 1
    if op == "s_nop":
        w = int(ops[0]) + 1
    elif not in_asm and (op.startswith("v_") or op.startswith("global_") or op.startswith("buffer_") or op.startswith("ds_") or op.startswith("flat_")):
        touched = set().union(*[regs(o.split()[0]) for o in ops if o]) if ops else set()
        for (dregs, left, text) in pend:
            hit = dregs & touched
            whole_c = op.startswith("v_mfma") and len(ops) >= 4 and regs(ops[3].split()[0]) == dregs and not (hit - regs(ops[3].split()[0]) - regs(ops[0].split()[0]))
            if hit and not whole_c:
                print("%s: compiler `%s` touches v%s with %d wait state(s) still owed by asm `%s`" % (kernel, t[:60], sorted(hit)[:4], left, text[:60]))
                bad += 1
    pend = [[d, left - w, text] for (d, left, text) in pend if left - w > 0]
    if in_asm and op.startswith("v_mfma"):
        pend.append([regs(ops[0].split()[0]), mfma_states(op), t])
    hist.append((op, regs(ops[0].split()[0]) if ops and ops[0] else set(), in_asm, w))
    hist = hist[-6:]
print("%d hazard(s)" % bad)
sys.exit(1 if bad else 0)

"""Lint of the hand-placed instruction streams (attn_pipe.hip): hipcc's hazard recogniser treats an `asm volatile` statement as opaque, so a
vector instruction IT emits right in front of an asm MFMA that reads the register gets no wait states (gfx950: VALU write -> MFMA source read
needs 2).  Compiles the file with -save-temps and reports every asm MFMA whose A / B / C operand overlaps the destination of a compiler-emitted
vector instruction fewer than 2 wait states ahead.  Exit code 1 when any is found.   python tools/lint_asm_hazards.py [file.hip]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "walkgpt_amd", "csrc", "attn_pipe.hip")
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffast-math", "-fno-finite-math-only", "-I",
                os.path.join(ROOT, "walkgpt_amd", "csrc"), "-save-temps=obj", "-c", src, "-o", os.path.join(tmp, "x.o")] + sys.argv[2:],
               check=True, capture_output=True, cwd=tmp)
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
bad, in_asm, hist, kernel = 0, False, [], "?"
for ln in open(os.path.join(tmp, asm)):
    t = ln.strip()
    if t.endswith(":") and t.startswith("_Z"):
        kernel = t[:60]
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
    hist.append((op, regs(ops[0].split()[0]) if ops and ops[0] else set(), in_asm, w))
    hist = hist[-6:]
print("%d hazard(s)" % bad)
sys.exit(1 if bad else 0)

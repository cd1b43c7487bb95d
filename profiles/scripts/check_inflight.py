"""Scan a gfx950 .s file: registers that an inline-asm `global_load_dwordx4 ... nt` writes must not be read (or copied to
AGPRs / scratch) before the next `s_waitcnt vmcnt(0)` -- the compiler does not track inline-asm loads, so a move of an
in-flight register would read stale data.  usage: check_inflight.py file.s [kernel substring]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
kern = None
inflight = {}
bad = 0
def regs(tok):
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m: return set((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r"([va])(\d+)$", tok)
    if m: return {(m.group(1), int(m.group(2)))}
    return set()
for ln, line in enumerate(src):
    t = line.strip()
    if t.endswith(":") and not t.startswith(".") and not t.startswith(";"):
        kern = t[:-1]; inflight = {}
    if want not in (kern or ""): continue
    if not t or t.startswith(";") or t.startswith("."): continue
    ops = re.split(r"[\s,]+", t)
    if ops[0] == "global_load_dwordx4" and t.endswith("nt"):
        for r in regs(ops[1]): inflight[r] = ln
        # address operand may not be in flight either
        for r in regs(ops[2]):
            if r in inflight and inflight[r] != ln: print("READ of in-flight %s at line %d: %s" % (r, ln, t)); bad += 1
        continue
    if ops[0] == "s_waitcnt" and "vmcnt(0)" in t:
        inflight = {}; continue
    if ops[0].startswith("s_") and ops[0] not in ("s_waitcnt",): 
        if ops[0] in ("s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_cbranch_execnz", "s_branch"): pass
        continue
    if ops[0] not in ("v_accvgpr_write_b32", "v_accvgpr_write", "scratch_store_dword", "scratch_store_dwordx2", "scratch_store_dwordx4",
                      "v_mov_b32", "v_mov_b64", "buffer_store_dword", "v_accvgpr_read_b32", "v_accvgpr_mov_b32"): continue
    for tok in ops[2:] if ops[0].startswith("v_") else ops[1:]:
        for r in regs(tok):
            if r in inflight:
                print("%s: COPY of in-flight %s (issued line %d) at line %d: %s" % (kern, r, inflight[r], ln, t)); bad += 1
print("violations:", bad)

import re, sys
lines = open(sys.argv[1]).read().split('\n')
pending = []  # list of sets of reg indices, in issue order
def regs_of(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    if m: return {int(m.group(1))}
    return set()
bad = 0
inasm = False
for i, l in enumerate(lines):
    t = l.strip()
    if t.startswith(';;#ASMSTART'): inasm = True; continue
    if t.startswith(';;#ASMEND'): inasm = False; continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    if inasm:
        if t.startswith('ds_read_b128'):
            dst = t.split()[1].rstrip(',')
            pending.append(regs_of(dst))
        elif t.startswith('ds_write'):
            pending.append(set())
        elif t.startswith('s_waitcnt lgkmcnt'):
            n = int(re.search(r'lgkmcnt\((\d+)\)', t).group(1))
            pending = pending[len(pending) - n:] if n else []
        continue
    # compiler instruction: does it touch pending regs?
    toks = re.findall(r'v\[\d+:\d+\]|v\d+', t)
    used = set()
    for tk in toks: used |= regs_of(tk)
    allp = set().union(*pending) if pending else set()
    if used & allp:
        bad += 1
        if bad <= 25: print(i + 1, t, sorted(used & allp)[:8])
    if t.startswith('s_waitcnt') and 'lgkmcnt(0)' in t: pending = []
print("hazards:", bad)

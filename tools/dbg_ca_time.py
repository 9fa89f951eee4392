import os, sys, subprocess
shape = sys.argv[1:4] if len(sys.argv) > 3 else ["8", "576", "196"]
for flags in [0, 1, 2, 4, 8, 16, 6, 14, 30, 29, 31]:
    env = dict(os.environ, LVQ_CA_DBG=str(flags), WARM="50")
    out = subprocess.run([sys.executable, "tools/stamps_ca.py"] + shape, env=env, capture_output=True, text=True).stdout
    ph = [l.split()[-3] for l in out.split("\n") if "done" in l]
    print(flags, "LN/A/B/C median us:", ph, flush=True)

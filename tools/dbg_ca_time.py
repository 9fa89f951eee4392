import os, sys, subprocess
shape = sys.argv[1:4] if len(sys.argv) > 3 else ["1", "32768", "196"]
for flags in [0, 2, 8, 10, 16]:
    env = dict(os.environ, LVQ_CA_DBG=str(flags), WARM="100")
    out = subprocess.run([sys.executable, "tools/stamps_ca.py"] + shape, env=env, capture_output=True, text=True).stdout
    ph = [l.split()[-3] for l in out.split("\n") if "done" in l]
    cyc = [l for l in out.split("\n") if l.startswith("phase A")]
    print(flags, "LN/A/B/C median us:", ph, cyc[0] if cyc else "", flush=True)

import os, sys, subprocess, json
for flags in [0, 1, 2, 4, 7, 8, 16, 24, 64, 128, 7+8+16, 7+8+16+64+128]:
    env = dict(os.environ, LVQ_CA_DBG=str(flags))
    out = subprocess.run([sys.executable, "tools/bench_ca.py", "--shapes", "headline,resampled_b8", "--modes", "mixed", "--iters", "10"], env=env, capture_output=True, text=True).stdout
    r = [json.loads(l) for l in out.strip().split("\n") if l.startswith("{")]
    print(flags, [list(x.values())[0]["mixed"]["fused_ms"] for x in r], flush=True)

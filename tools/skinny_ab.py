import sys, os
sys.argv = ["x", "default"]
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib.util
spec = importlib.util.spec_from_file_location("sb", os.path.join(ROOT, "tools", "skinny_bench.py"))
src = open(os.path.join(ROOT, "tools", "skinny_bench.py")).read().split("SHAPES =")[0]
exec(src)
for rep in range(2):
    run("fc1 ln=1", 2, 2048, 512, True, 0)
    run("fc1 ln=0", 2, 2048, 512, False, 0)
    run("qkv-like ln=1", 2, 1536, 512, True, 0)
    run("qkv-like ln=0", 2, 1536, 512, False, 0)
    run("out-proj", 1, 512, 512, False, 0)
    run("fc2", 1, 512, 2048, False, 0)
    run("fc2 mt2", 1, 512, 2048, False, 2 << 8)
    run("act N512 K2048", 2, 512, 2048, False, 0)

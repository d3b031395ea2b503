import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
r3d = importlib.import_module("3d_reconstruction_project_amd")
print(json.dumps(bench.bench_gicp(r3d, r3d.default_context(0), cpu=False)))

"""Wall time of GenerateMesh (64^3 density query + marching cubes + normals + colours + copy to the host) on a trained object:
   python tools/mesh_timing.py [res]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package(); ss = ge.load_tools()
sc = ss.make_scene(n_views=24, H=240, W=320, f=260.0, seed=1)
ds, obj = ge.make_problem(pkg, sc, {}); obj.set_backend(1)
obj.train(1500)
res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
obj.generate_mesh(res, 2.0)
t0 = time.perf_counter(); n = 20
for _ in range(n):
    nv, ni = obj.generate_mesh(res, 2.0)
dt = (time.perf_counter() - t0) / n
print("GenerateMesh res %d: %.3f ms per call, %d vertices, %d indices" % (res, 1e3 * dt, nv, ni))

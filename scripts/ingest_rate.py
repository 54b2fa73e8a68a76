"""OBJ ingest rate of the C++ host's LoadMeshFromFile, line by line against several threads.
Usage: python scripts/ingest_rate.py [triangles] [threads ...]   (writes a temporary OBJ file of the synthetic scene)"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpuraytracer_amd import host_py

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
threads = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8, 16]
sc = host_py.synthetic_scene(n)
v = sc["vertices"]
d = tempfile.mkdtemp()
path = os.path.join(d, "soup.obj")
t0 = time.perf_counter()
with open(path, "w") as f:
    pos, uv, nrm = v["position"], v["uv"], v["normal"]
    step = 300000
    for a in range(0, len(v), step):
        b = min(a + step, len(v))
        rows = []
        for i in range(a, b, 3):
            for k in range(3):
                p, t, q = pos[i + k], uv[i + k], nrm[i + k]
                rows.append("v %.6f %.6f %.6f\nvt %.6f %.6f\nvn %.6f %.6f %.6f\n" % (p[0], p[1], p[2], t[0], t[1], q[0], q[1], q[2]))
            rows.append("f -3/-3/-3 -2/-2/-2 -1/-1/-1\n")
        f.write("".join(rows))
size = os.path.getsize(path)
print("%d triangles, %.1f MB written in %.1f s" % (n, size / 1e6, time.perf_counter() - t0), flush=True)
lib = host_py.load()
base = None
for dedup in (0, 1):
    for t in threads:
        t0 = time.perf_counter()
        h = lib.rth_load_mesh_threads(path.encode(), dedup, t)
        dt = time.perf_counter() - t0
        assert h
        nv = lib.rth_mesh_vertex_count(h)
        lib.rth_mesh_free(h)
        base = dt if t == threads[0] else base
        print("dedup %d, %2d thread(s): %.2f s = %.0f MB/s, %.2f M triangles/s (x%.1f); %d vertices" % (dedup, t, dt, size / dt / 1e6, n / dt / 1e6, base / dt, nv), flush=True)
os.remove(path)
os.rmdir(d)

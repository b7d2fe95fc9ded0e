"""GPU probe: the lane count of rocRAND's HOST generator (librocrand, ROCRAND_RNG_PSEUDO_XORWOW, seed 0): where does lane 0's second draw appear?"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
orc = ge.load_oracle()
rr = C.CDLL("librocrand.so")
def gen(n, ordering=None):
    g = C.c_void_p(); assert rr.rocrand_create_generator(C.byref(g), 401) == 0
    if ordering is not None:
        assert rr.rocrand_set_ordering(g, ordering) == 0
    t = torch.zeros(n, dtype=torch.float32, device="cuda")
    assert rr.rocrand_generate_uniform(g, C.c_void_p(t.data_ptr()), C.c_size_t(n)) == 0
    torch.cuda.synchronize(); rr.rocrand_destroy_generator(g); return t.cpu().numpy()
d = np.zeros(4, np.uint32); orc.lib().orc_xorwow_lane_draws(C.c_uint64(0), 1, 0, 4, d.ctypes.data_as(C.c_void_p))
u = np.float32(2.3283064e-10) + d.astype(np.float32) * np.float32(2.3283064e-10)
for name, o in (("default", None), ("legacy(103)", 103), ("dynamic(104)", 104)):
    out = gen(1 << 23, o)
    print(name, "lane 0 draws 0..3 at", [np.nonzero(out == v)[0][:3].tolist() for v in u])
    n = 1 << 20
    a = gen(n, o); b = gen(3 * n + 12345, o)
    print(name, "prefix property", bool(np.array_equal(a, b[:n])))

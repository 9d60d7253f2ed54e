import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from chromegcn_amd import _lib
from tools.kbench import timeit
lib = _lib.load(); dev = torch.device("cuda")
d, S = int(sys.argv[1]), 2
for n in [int(a) for a in sys.argv[2:]]:
    x, h = torch.randn(S, n, d, device=dev), torch.randn(S, n, d, device=dev)
    W = torch.randn(d, d, device=dev) / d ** 0.5; b = torch.randn(d, device=dev) * 0.1
    wg = torch.randn(d, device=dev) / d ** 0.5; cg = torch.zeros(1, device=dev)
    xn, z = torch.empty_like(x), torch.empty_like(x); gate = torch.empty(S, n, device=dev)
    dummy = torch.zeros(4, dtype=torch.int32, device=dev)
    P = _lib.ptr; st = _lib.stream_ptr
    lib.cgcn_debug_set_fwd_split_bytes(0)
    run = lambda: lib.cgcn_layer_fwd(st(), n, S, d, P(dummy), P(dummy), None, None, P(x), P(W), P(b), P(wg), P(cg), P(xn), P(z), None, P(gate), 0.0, None, 1, P(h), None, 0, None)
    assert run() == 0; torch.cuda.synchronize()
    if os.environ.get("KT"):
        buf = np.zeros(8 * 16, dtype=np.uint64)
        raw = ctypes.CDLL(os.environ["CHROMEGCN_LIB"])
        assert raw.cgcn_debug_kt_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        t = buf.reshape(8, 16).astype(np.int64)
        for b_ in range(8):
            v = t[b_, 8:15]
            if v[0] == 0: continue
            print("wg", b_ * 32, "start->loads %.2f  (last tile:) top %.2f  T-written+bar %.2f  mfma %.2f  tanh %.2f  rowpass %.2f   total %.2f us" % (
                (v[1]-v[0])/100., (v[2]-v[0])/100., (v[3]-v[2])/100., (v[4]-v[3])/100., (v[5]-v[4])/100., (v[6]-v[5])/100., (v[6]-v[0])/100.))
    print(n, d, "dense_us", round(timeit(run, reps=100), 2))

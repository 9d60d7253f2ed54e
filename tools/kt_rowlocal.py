"""Phase timestamps of k_bwd_rowlocal (tuning tool).  Needs a -DKT_TIMING build loaded through CHROMEGCN_LIB:
  CGCN_EXTRA_FLAGS="-DKT_TIMING" python -c "from chromegcn_amd import _build; _build.build_library(out='variants/libcgcn_kt.so')"
  CHROMEGCN_LIB=$PWD/variants/libcgcn_kt.so python tools/kt_rowlocal.py [chromosome]
Stamps (100 MHz wall clock) of workgroups 0, 32, ... 224 of the LAST rowlocal launch of a train step (first layer's
backward), wave 0 (a dW wave): 0 entry, 1 setup done, 2 top of the workgroup's 4th tile, 3 its dW product done, 4 row pass of
the 5th tile done, 5 past the barrier, 6 loop done + partial written, 7 column sums written."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chromegcn_amd as C
from chromegcn_amd import synth
from chromegcn_amd.finetune import GCNStage


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "chr10"
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = C.ChromeGCN(128, 128, synth.N_LABELS, 0.2, True, 2).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(model, opt, "hic", dev, hip_graphs=False, input_grad=True, cache_input_aggregation=False)
    feats, hic = synth.synthetic_chromosome(name, d=128)
    stage.add_chromosome(name, feats, hic)
    for _ in range(5):
        stage.train_step(name)
    torch.cuda.synchronize()
    raw = ctypes.CDLL(os.environ["CHROMEGCN_LIB"])
    buf = np.zeros(8 * 16, dtype=np.uint64)
    assert raw.cgcn_debug_kt_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(8, 16).astype(np.int64)
    print(name, "n =", stage.chroms[name].n, "(k_bwd_rowlocal_rs, last row-local launch of the step = first layer; period KT_PERIOD)")
    for b in range(8):
        v = t[b]
        print("wg %3d  kernel %.2f us | row team wave 0: row pass %.2f  load issue %.2f  barrier wait %.2f   (period %.2f) | matrix team wave 8: dW %.2f  dHs %.2f  barrier wait %.2f  (period %.2f)" % (
            b * 32, (v[5] - v[4]) / 100., (v[1] - v[0]) / 100., (v[2] - v[1]) / 100., (v[3] - v[2]) / 100., (v[3] - v[0]) / 100.,
            (v[9] - v[8]) / 100., (v[10] - v[9]) / 100., (v[11] - v[10]) / 100., (v[11] - v[8]) / 100.))


if __name__ == "__main__":
    main()

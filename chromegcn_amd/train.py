"""Command-line entry of the GCN stage: what `python main.py ... -chrome_model gcn -adj_type hic ...`
(README.md:45, main.py:62-101) does after the window encoder's features have been saved -- build ChromeGCN,
optionally take the classifier head + BatchNorm affine from the pretrained CNN checkpoint (main.py:74-81), build the
optimizer (utils/util_methods.py:14-19) and run the epoch loop (runner.py:25-62) -- on the MI355X path.

    python -m chromegcn_amd.train -feat_dir <cnn_run_dir> -graph_root <graphs_dir> -hicsize 500000 -hicnorm SQRTVC \\
        -gcn_layers 2 -gcn_dropout 0.2 -optim sgd -lr 0.25 -epochs 1000 -model_name <out_dir> [-cnn_chkpt model.chkpt]

Inputs are the reference's own files: `<feat_dir>/chrom_feature_dict_{train,valid,test}.pt`
(utils/util_methods.py:183-199) and `<graph_root>/{split}_graphs_{hicsize}_{hicnorm}norm.pkl`
(data/7create_graph_new.py:197-202).  `-synthetic` replaces them by the seeded GM12878-shaped stand-in."""
from __future__ import annotations

import argparse
import os
import pickle
import sys
import time

import torch

from . import ChromeGCN, runner, synth


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-feat_dir", type=str, default=None)
    ap.add_argument("-graph_root", type=str, default=None)
    ap.add_argument("-hicsize", type=str, default="500000")          # config_args.py:47
    ap.add_argument("-hicnorm", type=str, default="SQRTVC")          # config_args.py:46
    ap.add_argument("-adj_type", type=str, default="hic", choices=["constant", "hic", "both", "none"])
    ap.add_argument("-gcn_layers", type=int, default=2)              # config_args.py:40
    ap.add_argument("-gcn_dropout", type=float, default=0.2)         # config_args.py:24
    ap.add_argument("-gate", action="store_true")                    # accepted, ignored (ChromeModels.py:22-31)
    ap.add_argument("-optim", type=str, default="sgd", choices=["adam", "sgd"])
    ap.add_argument("-lr", type=float, default=0.25)
    ap.add_argument("-lr_decay2", type=float, default=0)
    ap.add_argument("-epochs", type=int, default=100)
    ap.add_argument("-model_name", type=str, default="results/gcn_run")
    ap.add_argument("-cnn_chkpt", type=str, default=None, help="pretrained window-model checkpoint (main.py:74-81)")
    ap.add_argument("-load_gcn", type=str, default=None, help="ChromeGCN checkpoint to evaluate (main.py:66-69)")
    ap.add_argument("-synthetic", action="store_true")
    ap.add_argument("-synthetic_chroms", type=str, default=",".join(synth.HG19_LEN))
    ap.add_argument("-gpu_id", type=int, default=0)
    return ap.parse_args(argv)


def load_inputs(opt):
    data, graphs = {}, {}
    if opt.synthetic:
        for sp in ("train", "valid", "test"):
            data[sp], graphs[sp] = {}, {}
        for c in opt.synthetic_chroms.split(","):
            feats, hic = synth.synthetic_chromosome(c)
            data[synth.split_of(c)][c] = feats
            graphs[synth.split_of(c)][c] = hic
        return data, graphs
    if not opt.feat_dir:
        raise SystemExit("-feat_dir is required (or -synthetic)")
    for sp in ("train", "valid", "test"):
        data[sp] = torch.load(os.path.join(opt.feat_dir, "chrom_feature_dict_%s.pt" % sp), map_location="cpu")   # main.py:30-32
        graphs[sp] = None
        if opt.adj_type in ("hic", "both"):
            path = os.path.join(opt.graph_root, sp + "_graphs_" + opt.hicsize + "_" + opt.hicnorm + "norm.pkl")  # finetune.py:21
            with open(path, "rb") as f:
                graphs[sp] = pickle.load(f)
    return data, graphs


def main(argv=None):
    opt = parse(argv)
    if not torch.cuda.is_available():
        raise SystemExit("chromegcn_amd.train needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(opt.gpu_id)
    dev = torch.device("cuda", opt.gpu_id)
    data, graphs = load_inputs(opt)
    first = next(iter(data["train"].values())) if data["train"] else next(iter(data["test"].values()))
    d, n_class = first["forward"].shape[1], first["target"].shape[1]
    model = ChromeGCN(d, d, n_class, opt.gcn_dropout, opt.gate, opt.gcn_layers)       # main.py:62
    if opt.load_gcn:
        ck = torch.load(opt.load_gcn, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["model"])                                           # main.py:66-69
    elif opt.cnn_chkpt:
        sd = torch.load(opt.cnn_chkpt, map_location="cpu", weights_only=False)["model"]
        pick = lambda suffix: next(v for k, v in sd.items() if k.endswith(suffix))   # DataParallel prefixes (main.py:75)
        with torch.no_grad():                                                        # main.py:78-81
            model.out.weight.copy_(pick("model.classifier.weight")); model.out.bias.copy_(pick("model.classifier.bias"))
            model.batch_norm.weight.copy_(pick("model.batch_norm.weight")); model.batch_norm.bias.copy_(pick("model.batch_norm.bias"))
    model.to(dev)
    if opt.optim == "adam":                                                          # utils/util_methods.py:14-19
        optimizer = torch.optim.Adam(model.parameters(), betas=(0.9, 0.98), lr=opt.lr)
    else:
        optimizer = torch.optim.SGD(model.parameters(), lr=opt.lr, weight_decay=1e-6, momentum=0.9)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=100, gamma=0.5)  # main.py:86
    opt.test_only = bool(opt.load_gcn)
    opt.load_gcn = bool(opt.load_gcn)
    t0 = time.time()
    hist = runner.run_model(None, model, data["train"], data["valid"], data["test"], None, optimizer, scheduler, opt, None,
                            graphs=graphs)
    print("done: %d epochs in %.2f s -> %s" % (len(hist), time.time() - t0, opt.model_name))
    return hist


if __name__ == "__main__":
    main(sys.argv[1:])

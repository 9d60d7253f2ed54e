"""chromegcn_amd -- MI355X-native gated graph-convolution hot path of ChromeGCN.

Public surface mirrors the reference modules it replaces (see layers.py / graph.py / finetune.py)."""
from .graph import ChromGraph, HostCSR, normalize_graph, process_graph, upload, as_graph  # noqa: F401
from .handoff import FeatureCollector  # noqa: F401
from .layers import ChromeGCN, GraphConvolution  # noqa: F401

__all__ = ["ChromeGCN", "GraphConvolution", "ChromGraph", "HostCSR", "normalize_graph", "process_graph",
           "upload", "as_graph", "FeatureCollector"]

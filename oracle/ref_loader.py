"""Load the reference's model files from /root/reference (this container only).

Two modes:
  * genuine(): pure-torch files (dense score nets, SDE_dense/SDE_sparse, layers/common|node|edge
    _network_dense) imported by file path with the package __init__ files bypassed -- no stand-ins.
  * verbatim(): stand-in third-party packages (oracle/standins) + /root/reference on sys.path, then
    the normal `from Geom3D.models import ...`.
The reference never travels to the GPU box: every caller must skip when available() is False.
"""
import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("MSDE_REFERENCE_ROOT", "/root/reference")
_STANDINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "standins")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "Geom3D", "models", "MoleculeSDE"))


def genuine():
    """Return a namespace with the genuinely imported pure-torch reference classes."""
    assert available()
    base = os.path.join(REF_ROOT, "Geom3D", "models", "MoleculeSDE")
    if "refsde" not in sys.modules:
        pkg = types.ModuleType("refsde"); pkg.__path__ = [base]
        lay = types.ModuleType("refsde.layers"); lay.__path__ = [os.path.join(base, "layers")]
        sys.modules["refsde"] = pkg
        sys.modules["refsde.layers"] = lay

        def load(name, path):
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            return mod

        common = load("refsde.layers.common", os.path.join(base, "layers", "common.py"))
        node = load("refsde.layers.node_network_dense", os.path.join(base, "layers", "node_network_dense.py"))
        edge = load("refsde.layers.edge_network_dense", os.path.join(base, "layers", "edge_network_dense.py"))
        lay.MultiLayerPerceptron = common.MultiLayerPerceptron
        lay.NodeNetwork_dense = node.NodeNetwork_dense
        lay.NodeNetwork_dense_03 = node.NodeNetwork_dense_03
        lay.EdgeNetwork_dense = edge.EdgeNetwork_dense
        lay.EdgeNetwork_dense_03 = edge.EdgeNetwork_dense_03
        load("refsde.invariant_scorenetwork_dense", os.path.join(base, "invariant_scorenetwork_dense.py"))
        load("refsde.SDE_dense", os.path.join(base, "SDE_dense.py"))
        load("refsde.SDE_sparse", os.path.join(base, "SDE_sparse.py"))
    ns = types.SimpleNamespace()
    ns.MultiLayerPerceptron = sys.modules["refsde.layers.common"].MultiLayerPerceptron
    ns.NodeNetwork_dense = sys.modules["refsde.layers.node_network_dense"].NodeNetwork_dense
    ns.EdgeNetwork_dense = sys.modules["refsde.layers.edge_network_dense"].EdgeNetwork_dense
    ns.EdgeLayer = sys.modules["refsde.layers.edge_network_dense"].EdgeLayer
    ns.EdgeScoreNetwork_dense = sys.modules["refsde.invariant_scorenetwork_dense"].EdgeScoreNetwork_dense
    ns.NodeScoreNetwork_dense = sys.modules["refsde.invariant_scorenetwork_dense"].NodeScoreNetwork_dense
    ns.SDE_dense = sys.modules["refsde.SDE_dense"]
    ns.SDE_sparse = sys.modules["refsde.SDE_sparse"]
    return ns


def verbatim():
    """Import the reference packages on the stand-in third-party layer; returns a namespace."""
    assert available()
    if _STANDINS not in sys.path:
        sys.path.insert(0, _STANDINS)
    if REF_ROOT not in sys.path:
        sys.path.insert(1, REF_ROOT)
    models = importlib.import_module("Geom3D.models")
    sde = importlib.import_module("Geom3D.models.MoleculeSDE")
    ns = types.SimpleNamespace()
    ns.GNN = models.GNN
    ns.SchNet = models.SchNet
    ns.SDEModel2Dto3D_02 = sde.SDEModel2Dto3D_02
    ns.SDEModel3Dto2D_node_adj_dense = sde.SDEModel3Dto2D_node_adj_dense
    ns.sde2d3d = importlib.import_module("Geom3D.models.MoleculeSDE.SDE_model_2D_to_3D")
    ns.sde3d2d = importlib.import_module("Geom3D.models.MoleculeSDE.SDE_model_3D_to_2D_node_adj_dense")
    ns.SDE_sparse = importlib.import_module("Geom3D.models.MoleculeSDE.SDE_sparse")
    ns.SDE_dense = importlib.import_module("Geom3D.models.MoleculeSDE.SDE_dense")
    import torch_geometric.nn as tgnn
    import torch_geometric.utils as tgu
    import torch_scatter
    ns.standins = types.SimpleNamespace(nn=tgnn, utils=tgu, scatter=torch_scatter)
    return ns

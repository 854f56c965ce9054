"""Drop-in counterparts of `Geom3D.models` / `Geom3D.models.MoleculeSDE` for the MoleculeSDE pretrain
hot path (same class names, constructor arguments, forward signatures and state_dict keys)."""
from .gnn import GNN
from .schnet import SchNet
from .painn import PaiNN
from .sde_2d_to_3d import SDEModel2Dto3D_01, SDEModel2Dto3D_02
from .sde_3d_to_2d import SDEModel3Dto2D_node_adj_dense, SDEModel3Dto2D_node_adj_dense_02
from .nn import prepare_batch, CpuReplayNoise, DeviceNoise
from .sde import VESDE, VPSDE

__all__ = ["GNN", "SchNet", "PaiNN", "SDEModel2Dto3D_01", "SDEModel2Dto3D_02", "SDEModel3Dto2D_node_adj_dense", "SDEModel3Dto2D_node_adj_dense_02", "prepare_batch", "CpuReplayNoise", "DeviceNoise", "VESDE", "VPSDE"]

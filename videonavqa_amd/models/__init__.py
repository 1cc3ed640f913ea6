"""Drop-in model classes mirroring the reference's `models/` package (same class names,
constructor signatures and forward() contracts), computed by the HIP kernel library."""
from .film_attn_pt_stem import FiLMAttnPretrainedStem
from .film_global_pooling_pt_stem import FiLMGlobalPoolingPretrainedStem
from .time_multi_hop_pt_stem import TimeMultiHopFiLMPretrainedStem
from .obj_detector import ObjDetectCNN
from .q_only_lstm import QOnlyLSTM
from .v_only_cnn3d import VideoOnlyCNN3D
from .mac import MACNetwork

__all__ = ["FiLMAttnPretrainedStem", "FiLMGlobalPoolingPretrainedStem",
           "TimeMultiHopFiLMPretrainedStem", "ObjDetectCNN", "QOnlyLSTM", "VideoOnlyCNN3D", "MACNetwork"]

"""Build-container check (needs /root/reference): named_parameters() order of the drop-in classes == the reference's.
On a CUDA-less box the reference leaves film_layer unregistered, so those names are skipped there."""
import sys

sys.path.insert(0, "/root/reference")
sys.path.insert(0, ".")
from models.film_attn_pt_stem import FiLMAttnPretrainedStem as RA              # noqa: E402
from models.film_global_pooling_pt_stem import FiLMGlobalPoolingPretrainedStem as RG   # noqa: E402
from models.mac import MACNetwork as RM                                      # noqa: E402
from models.time_multi_hop_pt_stem import TimeMultiHopFiLMPretrainedStem as RT   # noqa: E402
import videonavqa_amd.models as M                                             # noqa: E402


def names(m):
    return [(n, tuple(p.shape)) for n, p in m.named_parameters() if p.requires_grad]


kw = dict(num_res_blocks=2, num_res_block_channels=8, num_input_channels=8)
ok = True
for R, P in ((RA, M.FiLMAttnPretrainedStem), (RG, M.FiLMGlobalPoolingPretrainedStem), (RT, M.TimeMultiHopFiLMPretrainedStem)):
    r, p = names(R(3, 12, 7, **kw)), names(P(3, 12, 7, **kw))
    p = [x for x in p if not x[0].startswith("film_layer")]
    print(R.__name__, r == p)
    ok &= r == p
r = names(RM(20, dim=16, embed_hidden=12, max_step=3, classes=7))
p = names(M.MACNetwork(20, dim=16, embed_hidden=12, max_step=3, classes=7))
print("MACNetwork", r == p)
sys.exit(0 if ok and r == p else 1)

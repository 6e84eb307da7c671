"""daam_amd -- the DAAM heat-map extraction path (castorini/daam v0.2.0: ``trace`` /
``DiffusionHeatMapHooker`` / ``compute_global_heat_map``) as hand-written HIP for MI355X
(gfx950) behind the reference's API.  See DESIGN.md."""
from .hook import *          # noqa: F401,F403
from .utils import *         # noqa: F401,F403
from .heatmap import *       # noqa: F401,F403
from .trace import *         # noqa: F401,F403
from .experiment import *    # noqa: F401,F403
from .evaluate import compute_iou, compute_ioa, load_mask, MeanEvaluator, UnsupervisedEvaluator  # noqa: F401

__version__ = '0.1.0'

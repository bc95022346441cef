from smirk_amd.smirk_encoder import *  # noqa: F401,F403  (SmirkEncoder, PoseEncoder, ShapeEncoder, ExpressionEncoder, create_backbone)
from smirk_amd.smirk_encoder import SmirkEncoder, PoseEncoder, ShapeEncoder, ExpressionEncoder, create_backbone  # noqa: F401

from smirk_amd.FLAME import FLAME  # noqa: F401

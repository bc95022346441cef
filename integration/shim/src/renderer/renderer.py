from smirk_amd.renderer import Renderer  # noqa: F401

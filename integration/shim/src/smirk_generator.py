from smirk_amd.smirk_generator import SmirkGenerator, ResnetBlock  # noqa: F401

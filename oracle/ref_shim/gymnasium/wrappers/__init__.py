from . import env_checker  # noqa: F401

from ..core import Wrapper


class PassiveEnvChecker(Wrapper):
    pass

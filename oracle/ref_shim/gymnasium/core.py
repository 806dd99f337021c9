class Env:
    metadata = {}
    render_mode = None
    spec = None
    observation_space = None
    action_space = None

    @property
    def unwrapped(self):
        return self

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self._action_space = None
        self._observation_space = None

    def __getattr__(self, name):
        # only reached when normal lookup fails
        if name == "env":
            raise AttributeError(name)
        if name.startswith("_"):
            raise AttributeError(f"accessing private attribute '{name}' is prohibited")
        return getattr(self.env, name)

    @property
    def observation_space(self):
        if self.__dict__.get("_observation_space") is None:
            return self.env.observation_space
        return self._observation_space

    @observation_space.setter
    def observation_space(self, space):
        self._observation_space = space

    @property
    def action_space(self):
        if self.__dict__.get("_action_space") is None:
            return self.env.action_space
        return self._action_space

    @action_space.setter
    def action_space(self, space):
        self._action_space = space

    @property
    def render_mode(self):
        return self.env.render_mode

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action, **kwargs):
        return self.env.step(action, **kwargs)

    def render(self, *a, **k):
        return self.env.render(*a, **k)

    def close(self):
        return self.env.close()


class ObservationWrapper(Wrapper):
    pass


class ActionWrapper(Wrapper):
    pass


class RewardWrapper(Wrapper):
    pass

"""Per-environment restatement of the reference's `LLE` host class (python/lle/env/env.py, reward_strategy.py) on an
oracle world -- TEST INFRASTRUCTURE for tests/test_gpu_env.py, like everything that imports oracle/."""
import numpy as np

from oracle import observers as oo

EXIT, GEM, DIED = 0, 1, 2


class OracleLLE:
    def __init__(self, world, obs_type="layered", state_type="state", walkable_lasers=True, multi_objective=False, padding_size=0):
        self.w = world
        self.obs_type, self.state_type, self.padding_size = obs_type, state_type, padding_size
        self.walkable_lasers, self.multi_objective = walkable_lasers, multi_objective
        self.n_agents = world.n_agents
        self.n_arrived = self.n_deads = 0          # RewardStrategy.__post_init__ / reset (reward_strategy.py:32-44)
        self.done = False

    def _gen(self, name):
        w, p = self.w, self.padding_size
        return {"layered": lambda: oo.layered_observe(w), "flattened": lambda: oo.flattened_observe(w),
                "partial3x3": lambda: oo.partial_observe(w, 3), "partial5x5": lambda: oo.partial_observe(w, 5),
                "partial7x7": lambda: oo.partial_observe(w, 7), "state": lambda: oo.state_observe(w, False),
                "normalized-state": lambda: oo.state_observe(w, True), "perspective": lambda: oo.perspective_observe(w),
                "layered-padded": lambda: oo.layered_padded_observe(w, p), "layered-padded-1": lambda: oo.layered_padded_observe(w, 1),
                "layered-padded-2": lambda: oo.layered_padded_observe(w, 2), "layered-padded-3": lambda: oo.layered_padded_observe(w, 3)}[name]()

    def get_observation(self):
        return self._gen(self.obs_type)

    def get_state(self):                            # ObservationGenerator.get_state: observe()[0] (observations.py:110-111)
        return self._gen(self.state_type)[0]

    def available_actions(self):
        return oo.available_actions(self.w, self.walkable_lasers)

    def reset(self, colours=None):                  # env.py:189-203
        self.w.reset()
        self.n_arrived = self.n_deads = 0
        self.done = False
        if colours is not None:
            for l, c in enumerate(colours):
                self.w.set_source(l, colour=int(c))

    def compute_reward(self, events):               # reward_strategy.py:58-75 / 90-109
        if self.multi_objective:
            r = np.zeros(4, np.float32)
            for ty, _a in events:
                if ty == DIED:
                    r[2] += -1.0
                    self.n_deads += 1
                elif ty == GEM:
                    r[0] += 1.0
                elif ty == EXIT:
                    r[1] += 1.0
                    self.n_arrived += 1
            if r[2] != 0:
                d = r[2]
                r[:] = 0
                r[2] = d
            elif self.n_arrived == self.n_agents:
                r[3] += 1.0
            return r
        reward = 0.0
        for ty, _a in events:
            if ty == DIED:
                reward += -1.0
                self.n_deads += 1
            elif ty == GEM:
                reward += 1.0
            elif ty == EXIT:
                reward += 1.0
                self.n_arrived += 1
        if self.n_arrived == self.n_agents:        # death_reward is never assigned in the reference (:60,71-72)
            reward += 1.0
        return np.array([reward], np.float32)

    def set_state(self, positions, gems, alive):   # LLE.set_state, env.py:208-217
        self.n_arrived = self.n_deads = 0
        events = self.w.set_state(positions, gems, alive)
        self.compute_reward(events)
        self.done = self.n_arrived == self.n_agents or self.n_deads > 0

    def metrics(self):                              # the per-agent entries of Step.info, env.py:174-176
        return {"has-arrived": [bool(x) for x in self.w.arrived()], "is-alive": [bool(x) for x in self.w.alive()]}

    def step(self, actions):                        # env.py:165-187
        assert not self.done or getattr(self, "free_running", False), "Cannot step in a done environment"
        events = self.w.step([int(a) for a in actions])
        reward = self.compute_reward(events)
        self.done = self.n_arrived == self.n_agents or self.n_deads > 0   # env.py:253-254
        return reward, self.done

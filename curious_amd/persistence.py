"""Logs and weight files of the agent in the reference's shapes (DDPG.logs / save_weights / load_weights,
baselines/her/ddpg.py:469-509).  Mixed into curious_amd.ddpg.DDPG."""
import pickle

import numpy as np
import torch


class PersistenceMixin:
    def logs(self, prefix=''):
        logs = []
        logs += [('stats_o/mean', float(self.o_stats.mean.mean()))]
        logs += [('stats_o/std', float(self.o_stats.std.mean()))]
        logs += [('stats_g/mean', float(self.g_stats.mean.mean()))]
        logs += [('stats_g/std', float(self.g_stats.std.mean()))]
        if prefix != '' and not prefix.endswith('/'):
            return [(prefix + '/' + key, val) for key, val in logs]
        return logs

    def _net_arrays(self, vec, critic):
        off = 0 if critic else self.off_pi
        flat = vec[off:off + (self.P_Q if critic else self.P_pi)].cpu().numpy()
        out, o = [], 0
        for s in self._shapes(critic):
            n = int(np.prod(s))
            out.append(flat[o:o + n].reshape(s).copy())
            o += n
        return out

    def _load_net_arrays(self, vec, critic, arrays):
        off = 0 if critic else self.off_pi
        flat = np.concatenate([np.asarray(a, dtype=np.float32).reshape(-1) for a in arrays])
        assert flat.size == (self.P_Q if critic else self.P_pi)
        vec[off:off + flat.size].copy_(torch.from_numpy(flat))

    def _stats_arrays(self, nz):
        d, s = nz.size, nz.state.cpu().numpy()
        # TF global-variable creation order of Normalizer (normalizer.py:31-45): sum, sumsq, count, mean, std
        return [s[:d].copy(), s[d:2 * d].copy(), s[2 * d:2 * d + 1].copy(), s[2 * d + 1:3 * d + 1].copy(),
                s[3 * d + 1:].copy()]

    def save_weights(self, path):
        """Pickled list of lists in the reference's order: main/Q, main/pi, target/Q, target/pi, o_stats, g_stats
        (ddpg.py:481-497)."""
        with open(path + '_weights.pkl', 'wb') as f:
            pickle.dump(self._weights_lists(), f)

    def _weights_lists(self):
        return [self._net_arrays(self.theta, True), self._net_arrays(self.theta, False),
                self._net_arrays(self.theta_target, True), self._net_arrays(self.theta_target, False),
                self._stats_arrays(self.o_stats), self._stats_arrays(self.g_stats)]

    def _set_weights_lists(self, weights):
        assert len(weights) == 6, 'expected main/Q, main/pi, target/Q, target/pi, o_stats, g_stats (ddpg.py:483-484)'
        self._load_net_arrays(self.theta, True, weights[0])
        self._load_net_arrays(self.theta, False, weights[1])
        self._load_net_arrays(self.theta_target, True, weights[2])
        self._load_net_arrays(self.theta_target, False, weights[3])
        for nz, arrs in ((self.o_stats, weights[4]), (self.g_stats, weights[5])):
            nz.state.copy_(torch.from_numpy(np.concatenate([np.asarray(a, np.float32).reshape(-1) for a in arrs])))

    def load_weights(self, path):
        with open(path + '_weights.pkl', 'rb') as f:
            weights = pickle.load(f)                                 # ddpg.py:499-509
        self._set_weights_lists(weights)

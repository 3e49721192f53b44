"""Actor/critic MLPs, DDPG losses and flat gradients (oracle side).  TEST INFRASTRUCTURE ONLY.

The TensorFlow-1 graph code of the reference cannot be imported in this container, so
this file *restates* it from source:
  * nn_modular_her            baselines/her/util.py:73-107   (state branch + bias-free goal branch)
  * nn                        baselines/her/util.py:56-71    (flat)
  * MultiTaskActorCritic      baselines/her/actor_critic.py:51-98
  * ActorCritic               baselines/her/actor_critic.py:5-48
  * losses / gradients        baselines/her/ddpg.py:436-449
  * flatten_grads             baselines/her/util.py:49-53
Checked by float64 finite differences and an independent torch-autograd implementation
(tests/test_oracle_networks.py); not pinned by reference vectors ("restated from source").

Parameter layout of one network = TF variable creation order (util.py:79-101):
  modular: _0_state/kernel[in,H], _0_state/bias[H], _0_goal/kernel[G,H],
           _1/kernel[H,H], _1/bias[H], _2/kernel, _2/bias, _3/kernel[H,out], _3/bias[out]
  flat:    _0/kernel[in,H], _0/bias[H], _1/..., _3/kernel[H,out], _3/bias[out]
Kernels are [in, out] row-major.  The flat vector of a DDPG agent is [theta_Q | theta_pi]
(ddpg.py:456 main_vars = main/Q + main/pi).
"""
import numpy as np


def net_shapes(dim_state_in, dim_goal, hidden, layers, dim_out, modular=True):
    """Shapes in flat order.  dim_state_in excludes the goal for modular nets."""
    shapes = []
    if modular:
        shapes += [(dim_state_in, hidden), (hidden,), (dim_goal, hidden)]
    else:
        shapes += [(dim_state_in + dim_goal, hidden), (hidden,)]
    for _ in range(layers - 1):
        shapes += [(hidden, hidden), (hidden,)]
    shapes += [(hidden, dim_out), (dim_out,)]
    return shapes


def numel(shapes):
    return int(sum(int(np.prod(s)) for s in shapes))


def xavier_init(shapes, rng, dtype=np.float32):
    """tf.contrib.layers.xavier_initializer (uniform, limit sqrt(6/(fan_in+fan_out))), zero biases
    (util.py:81,87-88,99).  The random stream itself is NOT TensorFlow's; only the distribution is."""
    out = []
    for s in shapes:
        if len(s) == 2:
            lim = np.sqrt(6.0 / (s[0] + s[1]))
            out.append(rng.uniform(-lim, lim, size=s).astype(dtype))
        else:
            out.append(np.zeros(s, dtype))
    return out


def flatten(arrs):
    return np.concatenate([a.reshape(-1) for a in arrs])


def unflatten(flat, shapes):
    out, off = [], 0
    for s in shapes:
        n = int(np.prod(s))
        out.append(flat[off:off + n].reshape(s))
        off += n
    return out


# ------------------------------------------------------------------ forward / backward
def mlp_forward(params, x_state, x_goal, modular=True, keep_pre=False):
    """Returns (out, cache).  Hidden layers ReLU, last layer linear (util.py:76,96).
    keep_pre (tests: which units sit on the edge of their ReLU): the cache also holds, per hidden layer, the pre-activations
    `pres` and the sum of the magnitudes of the terms each one is made of, `mags` -- what a float32 evaluation's rounding
    error scales with."""
    if modular:
        Ws, bs, Wg = params[0], params[1], params[2]
        rest = params[3:]
        pre = x_state @ Ws + bs + x_goal @ Wg                       # util.py:79-91
        x0 = x_state
        mag = (np.abs(x_state) @ np.abs(Ws) + np.abs(bs) + np.abs(x_goal) @ np.abs(Wg)) if keep_pre else None
    else:
        W0, b0 = params[0], params[1]
        rest = params[2:]
        x0 = x_state                    # flat: caller passes the full concat (actor_critic.py:35,43,46)
        pre = x0 @ W0 + b0
        mag = (np.abs(x0) @ np.abs(W0) + np.abs(b0)) if keep_pre else None
    pres, mags = [pre], [mag]
    h = np.maximum(pre, 0)
    acts = [h]
    nl = len(rest) // 2
    for i in range(nl):
        W, b = rest[2 * i], rest[2 * i + 1]
        pre = h @ W + b
        if i < nl - 1:
            if keep_pre:
                pres.append(pre)
                mags.append(np.abs(h) @ np.abs(W) + np.abs(b))
            h = np.maximum(pre, 0)
            acts.append(h)
    cache = dict(x0=x0, x_goal=x_goal, acts=acts, modular=modular)
    if keep_pre:
        cache.update(pres=pres, mags=mags)
    return pre, cache


def mlp_backward(params, cache, dout, flip=()):
    """Returns (grads list in param order, d x_state_or_concat).  Goal input grad is not needed.
    flip (tests): [(layer, row, unit)] whose relu' is taken on the OTHER side -- what a float32 evaluation does when the
    unit's pre-activation lies within its rounding error of zero."""
    modular = cache['modular']
    rest = params[3:] if modular else params[2:]
    nl = len(rest) // 2
    acts = cache['acts']
    grest = [None] * len(rest)
    d = dout
    for i in reversed(range(nl)):
        W = rest[2 * i]
        h_in = acts[i]
        grest[2 * i] = h_in.T @ d
        grest[2 * i + 1] = d.sum(axis=0)
        gate = h_in > 0
        for (L, r, c) in flip:
            if L == i:
                gate[r, c] = not gate[r, c]
        d = (d @ W.T) * gate
    if modular:
        g = [cache['x0'].T @ d, d.sum(axis=0), cache['x_goal'].T @ d]
        dx = d @ params[0].T
    else:
        g = [cache['x0'].T @ d, d.sum(axis=0)]
        dx = d @ params[0].T
    return g + grest, dx


class DDPGMath:
    """Losses and flat gradients of one DDPG agent (ddpg.py:419-449)."""

    def __init__(self, dimo, dimg, dimu, dimtd, hidden=256, layers=3, max_u=1., gamma=0.98,
                 clip_return=50., clip_pos_returns=True, action_l2=1., modular=True, dtype=np.float32):
        self.dimo, self.dimg, self.dimu, self.dimtd = dimo, dimg, dimu, dimtd
        self.hidden, self.layers, self.max_u = hidden, layers, max_u
        self.gamma, self.clip_return, self.clip_pos_returns = gamma, clip_return, clip_pos_returns
        self.action_l2, self.modular, self.dtype = action_l2, modular, dtype
        sd = dimo + (dimtd if modular else 0)
        self.pi_shapes = net_shapes(sd, dimg, hidden, layers, dimu, modular)
        self.Q_shapes = net_shapes(sd + dimu, dimg, hidden, layers, 1, modular)
        self.P_pi, self.P_Q = numel(self.pi_shapes), numel(self.Q_shapes)

    def init(self, rng):
        Q = xavier_init(self.Q_shapes, rng, self.dtype)
        pi = xavier_init(self.pi_shapes, rng, self.dtype)
        return np.concatenate([flatten(Q), flatten(pi)]).astype(self.dtype)

    def split(self, theta):
        Q = unflatten(theta[:self.P_Q], self.Q_shapes)
        pi = unflatten(theta[self.P_Q:self.P_Q + self.P_pi], self.pi_shapes)
        return Q, pi

    def _state(self, o, td):
        return np.concatenate([o, td], axis=1) if self.modular else o

    def actor(self, pi_params, o, td, g, keep_pre=False):
        x = self._state(o, td) if self.modular else np.concatenate([o, g], axis=1)
        z, cache = mlp_forward(pi_params, x, g, self.modular, keep_pre)
        pi = self.max_u * np.tanh(z)                                 # actor_critic.py:89
        return pi, z, cache

    def critic(self, Q_params, o, td, g, u_scaled, keep_pre=False):
        if self.modular:
            x = np.concatenate([o, td, u_scaled], axis=1)            # actor_critic.py:93,96
        else:
            x = np.concatenate([o, g, u_scaled], axis=1)             # actor_critic.py:43,46
        return mlp_forward(Q_params, x, g, self.modular, keep_pre)

    def losses_and_grads(self, theta_main, theta_target, batch, keep_pre=False, flip=None):
        """batch: dict o,g,u,task_descr,o_2,g_2,r (already clipped, ddpg.py:350-353).
        Returns dict(Q_loss, pi_loss, Q_pi[B,1], Q_grad[P_Q], pi_grad[P_pi], ...).
        keep_pre: + `edge` = {'Q': [...], 'pi': [...], 'Q_pi': [...]}: per hidden layer of the passes whose ReLU gates a
        gradient (critic(u), actor, critic(pi)) the array |pre-activation| / (sum of the magnitudes of its terms), [B, H].
        flip: (pass name, layer, row, unit), or a list of them: those units' relu' taken on the other side in that backward
        pass (mlp_backward)."""
        dt = self.dtype
        o, g, u, td = (batch[k].astype(dt) for k in ('o', 'g', 'u', 'task_descr'))
        o2, g2, r = (batch[k].astype(dt) for k in ('o_2', 'g_2', 'r'))
        r = r.reshape(-1, 1)
        B = o.shape[0]
        Qm, pim = self.split(theta_main)
        Qt, pit = self.split(theta_target)
        mu = dt(self.max_u)
        # target network on (o_2, g_2)  (ddpg.py:424-431)
        pi_t, _, _ = self.actor(pit, o2, td, g2)
        Q_t_pi, _ = self.critic(Qt, o2, td, g2, pi_t / mu)
        hi = dt(0.) if self.clip_pos_returns else dt(np.inf)
        target = np.clip(r + dt(self.gamma) * Q_t_pi, dt(-self.clip_return), hi)   # ddpg.py:437-438
        # main network on (o, g)
        pi, z, cache_pi = self.actor(pim, o, td, g, keep_pre)
        Q_pi, cache_Qpi = self.critic(Qm, o, td, g, pi / mu, keep_pre)
        Q, cache_Q = self.critic(Qm, o, td, g, u / mu, keep_pre)
        diff = target - Q
        Q_loss = np.mean(np.square(diff))                            # ddpg.py:439
        pi_loss = -np.mean(Q_pi) + dt(self.action_l2) * np.mean(np.square(pi / mu))  # ddpg.py:440-441
        # critic gradient wrt main/Q (target is stop_gradient)
        dQ = (dt(-2.0) / dt(B)) * diff
        fl = {'Q': [], 'pi': [], 'Q_pi': []}
        if flip is not None:
            for f in ([flip] if isinstance(flip[0], str) else flip):     # one (pass, layer, row, unit) or a list of them
                fl[f[0]].append(tuple(f[1:]))
        gQ, _ = mlp_backward(Qm, cache_Q, dQ, fl['Q'])
        # actor gradient wrt main/pi: through the critic input slot that holds pi/max_u
        dQpi = np.full_like(Q_pi, dt(-1.0) / dt(B))
        _, dx = mlp_backward(Qm, cache_Qpi, dQpi, fl['Q_pi'])
        sd = self.dimo + (self.dimtd if self.modular else self.dimg)
        d_pi_scaled = dx[:, sd:sd + self.dimu]
        dpi = d_pi_scaled / mu + dt(self.action_l2) * dt(2.0) * pi / (mu * mu * dt(B * self.dimu))
        dz = dpi * mu * (dt(1.0) - np.square(np.tanh(z)))
        gpi, _ = mlp_backward(pim, cache_pi, dz, fl['pi'])
        out = dict(Q_loss=Q_loss, pi_loss=pi_loss, Q_pi=Q_pi, Q=Q, pi=pi, target=target,
                   Q_grad=flatten(gQ).astype(dt), pi_grad=flatten(gpi).astype(dt))
        if keep_pre:
            out['edge'] = {name: [np.abs(p) / np.maximum(m, 1e-300) for p, m in zip(c['pres'], c['mags'])]
                           for name, c in (('Q', cache_Q), ('pi', cache_pi), ('Q_pi', cache_Qpi))}
        return out

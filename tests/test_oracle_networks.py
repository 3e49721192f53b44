"""oracle.networks (restated from the TF graph source) against finite differences and an
independent torch-autograd implementation of the same formulas."""
import numpy as np
import torch

from oracle.networks import DDPGMath


def _batch(rng, B, dimo, dimg, dimu, nb):
    td = np.eye(nb)[rng.randint(nb, size=B)]
    return dict(o=rng.randn(B, dimo), g=rng.randn(B, dimg), u=rng.uniform(-1, 1, (B, dimu)), task_descr=td,
                o_2=rng.randn(B, dimo), g_2=rng.randn(B, dimg), r=-(rng.rand(B, 1) > 0.5).astype(float))


def _torch_losses(m, theta, theta_t, batch):
    """Independent implementation with autograd (actor_critic.py:51-98, ddpg.py:436-449)."""
    th = torch.tensor(theta, dtype=torch.float64, requires_grad=True)
    tt = torch.tensor(theta_t, dtype=torch.float64)

    def split(v, shapes):
        out, off = [], 0
        for s in shapes:
            n = int(np.prod(s))
            out.append(v[off:off + n].reshape(s))
            off += n
        return out

    def net(p, xs, xg):
        h = torch.relu(xs @ p[0] + p[1] + xg @ p[2])
        h = torch.relu(h @ p[3] + p[4])
        h = torch.relu(h @ p[5] + p[6])
        return h @ p[7] + p[8]

    b = {k: torch.tensor(v, dtype=torch.float64) for k, v in batch.items()}
    Qm, pim = split(th[:m.P_Q], m.Q_shapes), split(th[m.P_Q:], m.pi_shapes)
    Qt, pit = split(tt[:m.P_Q], m.Q_shapes), split(tt[m.P_Q:], m.pi_shapes)
    pi_t = m.max_u * torch.tanh(net(pit, torch.cat([b['o_2'], b['task_descr']], 1), b['g_2']))
    Q_t = net(Qt, torch.cat([b['o_2'], b['task_descr'], pi_t / m.max_u], 1), b['g_2'])
    target = torch.clamp(b['r'] + m.gamma * Q_t, -m.clip_return, 0.).detach()
    pi = m.max_u * torch.tanh(net(pim, torch.cat([b['o'], b['task_descr']], 1), b['g']))
    Q_pi = net(Qm, torch.cat([b['o'], b['task_descr'], pi / m.max_u], 1), b['g'])
    Q = net(Qm, torch.cat([b['o'], b['task_descr'], b['u'] / m.max_u], 1), b['g'])
    Q_loss = torch.mean((target - Q) ** 2)
    pi_loss = -torch.mean(Q_pi) + m.action_l2 * torch.mean((pi / m.max_u) ** 2)
    gQ = torch.autograd.grad(Q_loss, th, retain_graph=True)[0][:m.P_Q]
    gpi = torch.autograd.grad(pi_loss, th)[0][m.P_Q:]
    return Q_loss.item(), pi_loss.item(), gQ.numpy(), gpi.numpy(), Q_pi.detach().numpy()


def test_losses_and_grads_match_torch_autograd_f64():
    rng = np.random.RandomState(0)
    m = DDPGMath(40, 12, 4, 4, hidden=32, layers=3, max_u=1.5, dtype=np.float64)
    theta, theta_t = m.init(rng), m.init(rng)
    batch = _batch(rng, 16, 40, 12, 4, 4)
    out = m.losses_and_grads(theta, theta_t, batch)
    Ql, pl, gQ, gpi, Qpi = _torch_losses(m, theta, theta_t, batch)
    np.testing.assert_allclose(out['Q_loss'], Ql, rtol=1e-12)
    np.testing.assert_allclose(out['pi_loss'], pl, rtol=1e-12)
    np.testing.assert_allclose(out['Q_grad'], gQ, rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out['pi_grad'], gpi, rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out['Q_pi'], Qpi, rtol=1e-12)


def test_finite_differences_f64():
    rng = np.random.RandomState(1)
    m = DDPGMath(10, 6, 4, 2, hidden=8, layers=3, dtype=np.float64)
    theta, theta_t = m.init(rng), m.init(rng)
    batch = _batch(rng, 6, 10, 6, 4, 2)
    out = m.losses_and_grads(theta, theta_t, batch)
    eps = 1e-6
    for i in rng.choice(m.P_Q, 25, replace=False):
        d = np.zeros_like(theta)
        d[i] = eps
        fd = (m.losses_and_grads(theta + d, theta_t, batch)['Q_loss']
              - m.losses_and_grads(theta - d, theta_t, batch)['Q_loss']) / (2 * eps)
        assert abs(fd - out['Q_grad'][i]) < 1e-8
    for i in rng.choice(m.P_pi, 25, replace=False):
        d = np.zeros_like(theta)
        d[m.P_Q + i] = eps
        fd = (m.losses_and_grads(theta + d, theta_t, batch)['pi_loss']
              - m.losses_and_grads(theta - d, theta_t, batch)['pi_loss']) / (2 * eps)
        assert abs(fd - out['pi_grad'][i]) < 1e-8


def test_param_counts_arm4():
    m = DDPGMath(40, 12, 4, 4)
    assert (m.P_pi, m.P_Q) == (147204, 147457)


def test_f32_close_to_f64():
    rng = np.random.RandomState(2)
    m64 = DDPGMath(40, 12, 4, 4, dtype=np.float64)
    m32 = DDPGMath(40, 12, 4, 4, dtype=np.float32)
    theta, theta_t = m64.init(rng), m64.init(rng)
    batch = _batch(rng, 256, 40, 12, 4, 4)
    a = m64.losses_and_grads(theta, theta_t, batch)
    b = m32.losses_and_grads(theta.astype(np.float32), theta_t.astype(np.float32), batch)
    assert abs(a['Q_loss'] - b['Q_loss']) <= 1e-5 * abs(a['Q_loss'])
    assert abs(a['pi_loss'] - b['pi_loss']) <= 1e-5 * abs(a['pi_loss'])
    np.testing.assert_allclose(b['Q_grad'], a['Q_grad'], rtol=0, atol=1e-5 * np.abs(a['Q_grad']).max())


def test_hand_case_H2_B1():
    """Hand-computable: H=2, one sample, all weights 0.5, inputs 1 (relu active everywhere)."""
    m = DDPGMath(1, 1, 1, 1, hidden=2, layers=1, max_u=1., gamma=0.5, clip_return=100., dtype=np.float64)
    theta = np.full(m.P_Q + m.P_pi, 0.5)
    batch = dict(o=np.ones((1, 1)), g=np.ones((1, 1)), u=np.ones((1, 1)), task_descr=np.ones((1, 1)),
                 o_2=np.ones((1, 1)), g_2=np.ones((1, 1)), r=-np.ones((1, 1)))
    out = m.losses_and_grads(theta, theta, batch)
    # actor: h = relu(0.5*2 + 0.5 + 0.5) = 2 (x2), z = 0.5*2*2 + 0.5 = 2.5, pi = tanh(2.5)
    pi = np.tanh(2.5)
    # critic(x): h = relu(0.5*(1+1+x) + 0.5 + 0.5) = 2 + 0.5x, Q = 0.5*2*h + 0.5 = 2.5 + 0.5x
    Q = 2.5 + 0.5 * 1.0
    Q_pi = 2.5 + 0.5 * pi
    target = np.clip(-1 + 0.5 * Q_pi, -100, 0)
    np.testing.assert_allclose(out['Q'][0, 0], Q)
    np.testing.assert_allclose(out['Q_pi'][0, 0], Q_pi)
    np.testing.assert_allclose(out['Q_loss'], (target - Q) ** 2)
    np.testing.assert_allclose(out['pi_loss'], -Q_pi + pi ** 2)

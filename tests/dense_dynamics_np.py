"""Independent fp64 formulation of the TOCABI free dynamics, used to pin the oracle's ABA.

Everything here is written in WORLD-frame spatial vectors about the world origin with dense body
Jacobians, M = sum_i J_i' I_i J_i, and a dense solve -- deliberately a different algorithm and a
different coordinate convention from oracle/dw_physics.c (body-frame recursive ABA), so agreement
between the two is a real check.  Test helper, never imported by the product.
"""
import numpy as np


def skew(p):
    return np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0.0]])


def quat_to_mat(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rodrigues(a, th):
    K = skew(a)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def crm(V):
    """spatial motion cross-product matrix"""
    w, v = V[:3], V[3:]
    out = np.zeros((6, 6))
    out[:3, :3] = skew(w)
    out[3:, :3] = skew(v)
    out[3:, 3:] = skew(w)
    return out


def crf(V):
    return -crm(V).T


class DenseDynamics:
    def __init__(self, model):
        self.m = model
        self.nb = 34
        self.parent = list(model.mv_parent)

    def kinematics(self, root_pos, root_quat, q):
        m = self.m
        Rw = [None] * self.nb
        pw = [None] * self.nb
        Rw[0] = quat_to_mat(np.asarray(root_quat, float) / np.linalg.norm(root_quat))
        pw[0] = np.asarray(root_pos, float)
        for b in range(1, self.nb):
            p = self.parent[b]
            R = np.array(m.mv_rot0[b]) @ rodrigues(np.array(m.mv_axis[b]), q[b - 1])
            Rw[b] = Rw[p] @ R
            pw[b] = pw[p] + Rw[p] @ np.array(m.mv_pos[b])
        return Rw, pw

    def terms(self, root_pos, root_quat, q, nu_b, qd, mass_scale, armature, damping, dt, g, tau):
        """Returns M (39x39), rhs (39), with generalised velocity [w_b, v_b (base coords), qd]."""
        m = self.m
        Rw, pw = self.kinematics(root_pos, root_quat, q)
        n = 6 + 33
        # base columns: body-coordinate twist -> world-origin twist
        X0 = np.zeros((6, 6))
        X0[:3, :3] = Rw[0]
        X0[3:, :3] = skew(pw[0]) @ Rw[0]
        X0[3:, 3:] = Rw[0]
        S = [None] * self.nb
        for b in range(1, self.nb):
            a = Rw[b] @ np.array(m.mv_axis[b])
            S[b] = np.concatenate([a, np.cross(pw[b], a)])
        nu = np.concatenate([nu_b, qd])
        J = [np.zeros((6, n)) for _ in range(self.nb)]
        V = [None] * self.nb
        Jdnu = [np.zeros(6) for _ in range(self.nb)]
        J[0][:, :6] = X0
        V[0] = X0 @ nu_b
        for b in range(1, self.nb):
            p = self.parent[b]
            J[b] = J[p].copy()
            J[b][:, 6 + b - 1] = S[b]
            V[b] = V[p] + S[b] * qd[b - 1]
            Jdnu[b] = Jdnu[p] + crm(V[b]) @ S[b] * qd[b - 1]
        # world-frame inertias
        I = [np.zeros((6, 6)) for _ in range(self.nb)]
        fg = [np.zeros(6) for _ in range(self.nb)]
        for k in range(36):
            b = m.inert_mv[k]
            ms = mass_scale[m.inert_gym[k]]
            mass = ms * m.inert_mass[k]
            c = pw[b] + Rw[b] @ np.array(m.inert_com[k])
            Ic = ms * (Rw[b] @ np.array(m.inert_I[k]) @ Rw[b].T)
            C = skew(c)
            Ik = np.zeros((6, 6))
            Ik[:3, :3] = Ic + mass * (C @ C.T)
            Ik[:3, 3:] = mass * C
            Ik[3:, :3] = mass * C.T
            Ik[3:, 3:] = mass * np.eye(3)
            I[b] += Ik
            fg[b] += np.concatenate([np.cross(c, mass * g), mass * g])
        M = np.zeros((n, n))
        h = np.zeros(n)
        for b in range(self.nb):
            M += J[b].T @ I[b] @ J[b]
            h += J[b].T @ (I[b] @ Jdnu[b] + crf(V[b]) @ (I[b] @ V[b]) - fg[b])
        Mrot = M.copy()
        for j in range(33):
            Mrot[6 + j, 6 + j] += armature[j]
        Meff = Mrot.copy()
        for j in range(33):
            Meff[6 + j, 6 + j] += dt * damping[j]
        rhs = -h
        rhs[6:] += tau - damping * qd
        return dict(M=Mrot, Meff=Meff, rhs=rhs, J=J, V=V, I=I, Rw=Rw, pw=pw, nu=nu)

    def accelerations(self, *a, **k):
        t = self.terms(*a, **k)
        acc = np.linalg.solve(t["Meff"], t["rhs"])
        return acc[:6], acc[6:], t

    def energy_momentum(self, root_pos, root_quat, q, nu_b, qd, mass_scale, armature, g):
        t = self.terms(root_pos, root_quat, q, nu_b, qd, mass_scale, armature, np.zeros(33), 0.0, g, np.zeros(33))
        ke = 0.5 * t["nu"] @ t["M"] @ t["nu"]
        pe = 0.0
        mom = np.zeros(6)
        m = self.m
        for b in range(self.nb):
            mom += t["I"][b] @ t["V"][b]
        for k in range(36):
            b = m.inert_mv[k]
            mass = mass_scale[m.inert_gym[k]] * m.inert_mass[k]
            c = t["pw"][b] + t["Rw"][b] @ np.array(m.inert_com[k])
            pe -= mass * g @ c
        return ke, pe, mom


def body_twist_from_root(root, com0, vel_at_com=True):
    """[N?]13 root state (Gym layout) -> base-coordinate twist [w_b, v_b]."""
    R = quat_to_mat(root[3:7] / np.linalg.norm(root[3:7]))
    ww = np.array(root[10:13], float)
    vo = np.array(root[7:10], float)
    if vel_at_com:
        vo = vo - np.cross(ww, R @ np.asarray(com0))
    return np.concatenate([R.T @ ww, R.T @ vo])

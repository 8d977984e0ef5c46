"""Test harness: the interior-point main loop of src/solver/core/solver.rs:242-434 restated in
Python over a pluggable backend -- the CPU oracle or the HIP C ABI -- so that the reference's
own END-TO-END known answers (tests/basic_qp.rs, basic_lp.rs, basic_socp.rs) pin the whole
L1-L3 chain: KKT assembly, LDL', refinement, cone scalings and step operations, RHS algebra,
residuals.  Zero / Nonnegative / SecondOrder / Exponential / Power cones (the nonsymmetric path
with its scaling-strategy checkpoints, solver.rs:586-665); no presolve and no
equilibration (data.equilibration = identity), which changes the iterates but not the optimum
the reference tests assert to 1e-3 ... 1e-6.

Test infrastructure only: the product is the C ABI underneath `HipBackend`.
"""
import math

import numpy as np

ZERO, NN, SOC, EXP, POW, GENPOW, PSD = 0, 1, 2, 3, 4, 5, 6
AFFINE, COMBINED = 0, 1
PRIMAL_DUAL, DUAL = 0, 1  # ScalingStrategy, core/solver.rs:77-80


class Vars:
    """DefaultVariables (default/variables.rs:12-50) on the host"""

    def __init__(self, n, m):
        self.x, self.s, self.z = np.zeros(n), np.zeros(m), np.zeros(m)
        self.tau, self.kappa = 1.0, 1.0


# ---------------------------------------------------------------------------------------------
class OracleBackend:
    def __init__(self, oracle, n, m, P, A, q, b, cones):
        self.n, self.m = n, m
        if any(c[0] == PSD for c in cones):  # PSD cones: numpy restatement (oracle/psd_numpy.py)
            from oracle import psd_numpy
            self.cones = psd_numpy.MixedCones(oracle, cones)
            self.ks = oracle.KKTSolver(n, m, P, A, self.cones.c)
            self.sys = psd_numpy.KKTSystemPy(oracle, self.ks, self.cones, n, m, P, A, q, b)
        else:
            self.cones = oracle.Cones(cones)
            self.ks = oracle.KKTSolver(n, m, P, A, self.cones)
            self.sys = oracle.KKTSystem(self.ks, self.cones, n, m, P, A, q, b)
        self.degree = self.cones.degree
        self.is_symmetric = self.cones.is_symmetric
        self.allows_primal_dual_scaling = not any(c[0] == GENPOW for c in cones)

    def update_scaling(self, s, z, mu, strategy):
        return self.cones.update_scaling(s, z, mu, strategy)

    def unit_initialization(self, z, s):
        self.cones.unit_initialization(z, s)

    def compute_barrier(self, z, s, dz, ds, alpha):
        return self.cones.compute_barrier(z, s, dz, ds, alpha)

    def kkt_update(self):
        return self.sys.update()

    def kkt_solve(self, lhs, rhs, variables, direction):
        return self.sys.solve(lhs, rhs, variables, direction)

    def solve_initial_point(self, variables):
        return self.sys.solve_initial_point(variables)

    def residuals(self, variables):
        return self.sys.residuals(variables)

    def affine_ds(self, s):
        return self.cones.affine_ds(s)

    def combined_ds_shift(self, step_z, step_s, sigma_mu):
        shift, wz, ws = self.cones.combined_ds_shift(step_z, step_s, sigma_mu)
        step_z[:], step_s[:] = wz, ws
        return shift

    def step_length(self, dz, ds, z, s, alpha_max):
        return self.cones.step_length(dz, ds, z, s, alpha_max)

    def margins(self, z):
        return self.cones.margins(z)

    def scaled_unit_shift(self, z, alpha, primal):
        self.cones.scaled_unit_shift(z, alpha, primal)


class HipBackend:
    """every operation runs on the device through the C ABI; host arrays are staged per call
    (this harness checks results, it is not a throughput path)"""

    def __init__(self, hip, n, m, P, A, q, b, cones):
        self.hip, self.n, self.m = hip, n, m
        Pm, Am = hip.CscMatrix(n, n, *P), hip.CscMatrix(m, n, *A)
        self.ks = hip.HipKKTSolver(Pm, Am, cones, m, n)
        self.sys = hip.HipKKTSystem(self.ks, Pm, Am, q, b)
        self.degree = sum(c[1] if c[0] in (NN, PSD) else (1 if c[0] == SOC else (3 if c[0] in (EXP, POW) else (
            c[1] + 1 if c[0] == GENPOW else 0))) for c in cones)
        self.is_symmetric = not any(c[0] in (EXP, POW, GENPOW) for c in cones)
        self.allows_primal_dual_scaling = not any(c[0] == GENPOW for c in cones)
        D = hip.DeviceArray
        self._v = [hip.DeviceVariables(n, m) for _ in range(3)]  # lhs, rhs, variables
        self._r = dict(rx=D(n), rz=D(m), rx_inf=D(n), rz_inf=D(m), Px=D(n))
        self._t = [D(m) for _ in range(5)]

    def _put(self, dv, v):
        dv.x.copy_from(v.x)
        dv.s.copy_from(v.s)
        dv.z.copy_from(v.z)
        dv.tau, dv.kappa = v.tau, v.kappa

    def update_scaling(self, s, z, mu, strategy):
        return self.ks.update_scaling(s, z, mu, strategy)

    def kkt_update(self):
        return self.sys.update()

    def kkt_solve(self, lhs, rhs, variables, direction):
        dl, dr, dv = self._v
        self._put(dr, rhs)
        self._put(dv, variables)
        ok = self.sys.solve(dl, dr, dv, direction)
        if ok:
            lhs.x[:], lhs.s[:], lhs.z[:] = dl.x.numpy(), dl.s.numpy(), dl.z.numpy()
            lhs.tau, lhs.kappa = dl.tau, dl.kappa
        return ok

    def solve_initial_point(self, variables):
        dv = self._v[2]
        self._put(dv, variables)
        ok = self.sys.solve_initial_point(dv)
        variables.x[:], variables.s[:], variables.z[:] = dv.x.numpy(), dv.s.numpy(), dv.z.numpy()
        return ok

    def residuals(self, variables):
        dv = self._v[2]
        self._put(dv, variables)
        r = self._r
        out = self.sys.residuals_update(dv, r["rx"], r["rz"], r["rx_inf"], r["rz_inf"], r["Px"])
        out.update({k: v.numpy() for k, v in r.items()})
        return out

    def affine_ds(self, s):
        t = self._t
        t[0].copy_from(s)
        self.ks.affine_ds_dev(t[1].ptr, t[0].ptr)
        return t[1].numpy()

    def combined_ds_shift(self, step_z, step_s, sigma_mu):
        t = self._t
        t[0].copy_from(step_z)
        t[1].copy_from(step_s)
        self.ks.combined_ds_shift_dev(t[2].ptr, t[0].ptr, t[1].ptr, sigma_mu)
        step_z[:], step_s[:] = t[0].numpy(), t[1].numpy()
        return t[2].numpy()

    def step_length(self, dz, ds, z, s, alpha_max):
        t = self._t
        for k, v in enumerate((dz, ds, z, s)):
            t[k].copy_from(v)
        return self.ks.step_length_dev(t[0].ptr, t[1].ptr, t[2].ptr, t[3].ptr, alpha_max)

    def margins(self, z):
        self._t[0].copy_from(z)
        return self.ks.margins_dev(self._t[0].ptr)

    def scaled_unit_shift(self, z, alpha, primal):
        self._t[0].copy_from(z)
        self.ks.scaled_unit_shift_dev(self._t[0].ptr, alpha, primal)
        z[:] = self._t[0].numpy()

    def unit_initialization(self, z, s):
        t = self._t
        self.ks.unit_initialization_dev(t[0].ptr, t[1].ptr)
        z[:], s[:] = t[0].numpy(), t[1].numpy()

    def compute_barrier(self, z, s, dz, ds, alpha):
        t = self._t
        for k, v in enumerate((z, s, dz, ds)):
            t[k].copy_from(v)
        return self.ks.compute_barrier_dev(t[0].ptr, t[1].ptr, t[2].ptr, t[3].ptr, alpha)


# ---------------------------------------------------------------------------------------------
def _shift_to_cone_interior(be, z, primal):
    # default/variables.rs:231-256
    min_margin, pos_margin = be.margins(z)
    target = max(1.0, (pos_margin * 0.1) / max(be.degree, 1))
    if min_margin <= 0.0:
        be.scaled_unit_shift(z, -min_margin, primal)
        be.scaled_unit_shift(z, target, primal)
    elif min_margin < target:
        be.scaled_unit_shift(z, target - min_margin, primal)
    else:
        be.scaled_unit_shift(z, 0.0, primal)


def _step_length(be, variables, step, direction, max_step_fraction):
    # default/variables.rs:114-156
    a_tau = -variables.tau / step.tau if step.tau < 0 else math.inf
    a_kap = -variables.kappa / step.kappa if step.kappa < 0 else math.inf
    alpha = min(a_tau, a_kap, 1.0)
    alpha = be.step_length(step.z, step.s, variables.z, variables.s, alpha)
    if direction == COMBINED:
        alpha *= max_step_fraction
    return alpha


def _barrier(be, variables, step, alpha):
    # default/variables.rs:186-207
    coef = be.degree + 1
    ctau, ckap = variables.tau + alpha * step.tau, variables.kappa + alpha * step.kappa
    sz = float(np.dot(variables.s + alpha * step.s, variables.z + alpha * step.z))
    mu = (sz + ctau * ckap) / coef
    lg = lambda v: -math.inf if v <= 0 else math.log(v)
    return coef * lg(mu) - lg(ctau) - lg(ckap) + be.compute_barrier(variables.z, variables.s, step.z, step.s, alpha)


def solve(be, cones, q, b, max_iter=200, tol_gap_abs=1e-8, tol_gap_rel=1e-8, tol_feas=1e-8,
          max_step_fraction=0.99, min_terminate_step_length=1e-4, min_switch_step_length=1e-1,
          linesearch_backtrack_step=0.8, trace=None):
    """-> dict(status, x, s, z, obj_val, iterations).  `trace`, if a list, receives
    (mu, alpha, sigma, res_primal, res_dual, gap_abs) per iteration for trajectory parity."""
    n, m = be.n, be.m
    q, b = np.asarray(q, float), np.asarray(b, float)
    variables, lhs, rhs = Vars(n, m), Vars(n, m), Vars(n, m)
    normq, normb = float(np.max(np.abs(q))) if n else 0.0, float(np.max(np.abs(b))) if m else 0.0
    symmetric = be.is_symmetric
    # default_start (solver.rs:525-543)
    if symmetric:
        e, e2 = np.zeros(m), np.zeros(m)
        be.unit_initialization(e, e2)  # the identity element: set_identity_scaling, compositecone.rs:216-222
        assert be.update_scaling(e, e, 1.0, 0)
        be.kkt_update()
        be.solve_initial_point(variables)
        _shift_to_cone_interior(be, variables.s, True)
        _shift_to_cone_interior(be, variables.z, False)
    else:
        be.unit_initialization(variables.z, variables.s)  # variables.rs:167-173
        variables.x[:] = 0.0
    variables.tau = variables.kappa = 1.0
    it, alpha, sigma = 0, 0.0, 1.0
    status = "Unsolved"
    scaling = PRIMAL_DUAL if be.allows_primal_dual_scaling else DUAL  # solver.rs:277-280
    while True:
        res = be.residuals(variables)
        mu = (res["dot_sz"] + variables.tau * variables.kappa) / (be.degree + 1)
        # default/info.rs:112-180 with identity equilibration
        tinv = 1.0 / variables.tau
        xPx2 = res["dot_xPx"] * tinv * tinv / 2.0
        cost_primal = res["dot_qx"] * tinv + xPx2
        cost_dual = -res["dot_bz"] * tinv - xPx2
        normx, normz, norms = (np.linalg.norm(v) * tinv for v in (variables.x, variables.z, variables.s))
        res_primal = np.linalg.norm(res["rz"]) * tinv / max(1.0, normb + normx + norms)
        res_dual = np.linalg.norm(res["rx"]) * tinv / max(1.0, normq + normx + normz)
        gap_abs = abs(cost_primal - cost_dual)
        gap_rel = gap_abs / max(1.0, min(abs(cost_primal), abs(cost_dual)))
        ktratio = variables.kappa * tinv
        if trace is not None:
            trace.append((mu, alpha, sigma, res_primal, res_dual, gap_abs))
        if ktratio <= 1.0 and (gap_abs < tol_gap_abs or gap_rel < tol_gap_rel) and res_primal < tol_feas \
                and res_dual < tol_feas:
            status = "Solved"
            break
        if it == max_iter:
            status = "MaxIterations"
            break
        if not be.update_scaling(variables.s, variables.z, mu, scaling):
            status = "NumericalError"
            break
        it += 1
        ok = be.kkt_update()
        # affine step (variables.rs:66-78)
        rhs.x[:], rhs.z[:] = res["rx"], res["rz"]
        rhs.s[:] = be.affine_ds(variables.s)
        rhs.tau, rhs.kappa = res["rtau"], variables.tau * variables.kappa
        ok = ok and be.kkt_solve(lhs, rhs, variables, AFFINE)
        if ok:
            alpha = _step_length(be, variables, lhs, AFFINE, max_step_fraction)
            sigma = (1.0 - alpha) ** 3
            mm = 1.0 if it > 1 else alpha
            # combined step (variables.rs:80-112)
            sm = sigma * mu
            rhs.x[:] = (1.0 - sigma) * res["rx"]
            rhs.tau = (1.0 - sigma) * res["rtau"]
            rhs.kappa = -sm + mm * lhs.tau * lhs.kappa + variables.tau * variables.kappa
            if mm != 1.0:
                lhs.z *= mm
            shift = be.combined_ds_shift(lhs.z, lhs.s, sm)
            rhs.s[:] = rhs.s + shift
            rhs.z[:] = (1.0 - sigma) * res["rz"]
            ok = be.kkt_solve(lhs, rhs, variables, COMBINED)
        if not ok:  # strategy_checkpoint_numerical_error, solver.rs:610-628
            if not symmetric and scaling == PRIMAL_DUAL:
                alpha, scaling = 0.0, DUAL
                continue
            status = "NumericalError"
            break
        alpha = _step_length(be, variables, lhs, COMBINED, max_step_fraction)
        if not symmetric and scaling == DUAL:  # backtrack_step_to_barrier, solver.rs:571-584
            for _ in range(50):
                if _barrier(be, variables, lhs, alpha) < 1.0:
                    break
                alpha *= linesearch_backtrack_step
        # strategy_checkpoint_small_step, solver.rs:630-650
        if not symmetric and scaling == PRIMAL_DUAL and alpha < min_switch_step_length:
            alpha, scaling = 0.0, DUAL
            continue
        if alpha <= max(0.0, min_terminate_step_length):
            status = "InsufficientProgress"
            break
        variables.x += alpha * lhs.x
        variables.s += alpha * lhs.s
        variables.z += alpha * lhs.z
        variables.tau += alpha * lhs.tau
        variables.kappa += alpha * lhs.kappa
    tinv = 1.0 / variables.tau
    return dict(status=status, x=variables.x * tinv, s=variables.s * tinv, z=variables.z * tinv,
                obj_val=cost_primal, iterations=it)

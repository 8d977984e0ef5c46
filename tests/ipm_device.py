"""Device-resident variant of tests/ipm_driver.py: the same restatement of the reference's main
loop (core/solver.rs:242-464, identity equilibration, no presolve), but x, s, z, the residuals and
both step vectors never leave HBM -- every vector operation goes through the L2/L3 C ABI
(chip_kktsystem_*, chip_residuals_update, chip_variables_*), only scalars cross the boundary.

Test / measurement harness (the IPM driver itself is out of scope); optional per-phase
wall-clock accounting for tools/ipm_scale.py."""
import math
import time

import numpy as np

NN, SOC, EXP, POW, GENPOW, PSD = 1, 2, 3, 4, 5, 6
PRIMAL_DUAL, DUAL = 0, 1
AFFINE, COMBINED = 0, 1


class _Clock:
    def __init__(self, sync):
        self.t, self.sync, self.on = {}, sync, False

    def __call__(self, key):
        return _Span(self, key)


class _Span:
    def __init__(self, clock, key):
        self.c, self.k = clock, key

    def __enter__(self):
        if self.c.on:
            self.c.sync()
            self.t0 = time.perf_counter()

    def __exit__(self, *a):
        if self.c.on:
            self.c.sync()
            self.c.t[self.k] = self.c.t.get(self.k, 0.0) + time.perf_counter() - self.t0


def solve_device(hip, n, m, P, A, q, b, cones, max_iter=200, tol_gap_abs=1e-8, tol_gap_rel=1e-8,
                 tol_feas=1e-8, max_step_fraction=0.99, min_terminate_step_length=1e-4,
                 min_switch_step_length=1e-1, linesearch_backtrack_step=0.8, settings=None, trace=None,
                 timing=None, fetch=True):
    """-> dict(status, x, s, z, obj_val, iterations[, timing]).  `timing`, if a dict, receives
    seconds per phase (each phase bracketed by stream synchronisations) and 'iter_s' (whole loop,
    unbracketed when timing is None)."""
    q, b = np.asarray(q, float), np.asarray(b, float)
    Pm, Am = hip.CscMatrix(n, n, *P), hip.CscMatrix(m, n, *A)
    ks = hip.HipKKTSolver(Pm, Am, cones, m, n, settings=settings)
    sysd = hip.HipKKTSystem(ks, Pm, Am, q, b)
    degree = ks.degree()
    symmetric = not any(c[0] in (EXP, POW, GENPOW) for c in cones)
    dual_only = any(c[0] == GENPOW for c in cones)
    D = hip.DeviceArray
    variables, lhs, rhs = (hip.DeviceVariables(n, m) for _ in range(3))
    rx, rz, rx_inf, rz_inf, Px = D(n), D(m), D(n), D(m), D(n)
    normq = float(np.max(np.abs(q))) if n else 0.0
    normb = float(np.max(np.abs(b))) if m else 0.0
    clock = _Clock(ks.synchronize)
    clock.on = timing is not None

    # default_start (solver.rs:525-543)
    if symmetric:
        e = D(m)
        ks.unit_initialization_dev(e.ptr, rz.ptr)  # identity scaling (compositecone.rs:216-222)
        assert ks.update_scaling_dev(e.ptr, e.ptr, 1.0, 0)
        sysd.update()
        sysd.solve_initial_point(variables)
        sysd.symmetric_initialization(variables)
    else:
        sysd.unit_initialization(variables)
    it, alpha, sigma = 0, 0.0, 1.0
    status = "Unsolved"
    scaling = DUAL if dual_only else PRIMAL_DUAL
    ks.synchronize()
    t_loop = time.perf_counter()
    while True:
        with clock("residuals+info"):
            res = sysd.residuals_update(variables, rx, rz, rx_inf, rz_inf, Px, norms=True)
            mu = sysd.calc_mu(variables, res["dot_sz"])
            tinv = 1.0 / variables.tau
            xPx2 = res["dot_xPx"] * tinv * tinv / 2.0
            cost_primal = res["dot_qx"] * tinv + xPx2
            cost_dual = -res["dot_bz"] * tinv - xPx2
            nx, nz, ns, nrz, nrx = res["norms"]
            normx, normz, norms = nx * tinv, nz * tinv, ns * tinv
            res_primal = nrz * tinv / max(1.0, normb + normx + norms)
            res_dual = nrx * tinv / max(1.0, normq + normx + normz)
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
        with clock("update_scaling"):
            ok = ks.update_scaling_dev(variables.s.ptr, variables.z.ptr, mu, scaling)
        if not ok:
            status = "NumericalError"
            break
        it += 1
        with clock("kkt_update"):
            ok = sysd.update()
        with clock("step_rhs"):
            sysd.affine_step_rhs(rhs, rx, rz, res["rtau"], variables)
        with clock("kkt_solve"):
            ok = ok and sysd.solve(lhs, rhs, variables, AFFINE)
        if ok:
            with clock("step_length"):
                alpha = sysd.calc_step_length(variables, lhs, AFFINE, max_step_fraction)
            sigma = (1.0 - alpha) ** 3
            mm = 1.0 if it > 1 else alpha
            with clock("step_rhs"):
                sysd.combined_step_rhs(rhs, rx, rz, res["rtau"], variables, lhs, sigma, mu, mm)
            with clock("kkt_solve"):
                ok = sysd.solve(lhs, rhs, variables, COMBINED)
        if not ok:  # strategy_checkpoint_numerical_error, solver.rs:610-628
            if not symmetric and scaling == PRIMAL_DUAL:
                alpha, scaling = 0.0, DUAL
                continue
            status = "NumericalError"
            break
        with clock("step_length"):
            alpha = sysd.calc_step_length(variables, lhs, COMBINED, max_step_fraction)
            if not symmetric and scaling == DUAL:  # backtrack_step_to_barrier, solver.rs:571-584
                for _ in range(50):
                    if sysd.barrier(variables, lhs, alpha) < 1.0:
                        break
                    alpha *= linesearch_backtrack_step
        if not symmetric and scaling == PRIMAL_DUAL and alpha < min_switch_step_length:
            alpha, scaling = 0.0, DUAL
            continue
        if alpha <= max(0.0, min_terminate_step_length):
            status = "InsufficientProgress"
            break
        with clock("add_step"):
            sysd.add_step(variables, lhs, alpha)
    ks.synchronize()
    t_loop = time.perf_counter() - t_loop
    tinv = 1.0 / variables.tau
    out = dict(status=status, obj_val=cost_primal, iterations=it, loop_s=t_loop, res_primal=res_primal,
               res_dual=res_dual, gap_abs=gap_abs, info=ks.linear_solver_info())
    if fetch:
        out.update(x=variables.x.numpy() * tinv, s=variables.s.numpy() * tinv, z=variables.z.numpy() * tinv)
    if timing is not None:
        timing.update(clock.t)
    return out

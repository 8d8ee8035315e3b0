"""Host-side state handling of the packed representation and the LM loop (-m "not gpu", TEST stand-in kernels):
in-place edits of variable tensors are seen, result buffers handed to the user are never recycled, Info bookkeeping."""
import numpy as np
import pytest
import torch

from tests.helpers import golden_problem, load_golden


def _layer(name="pg_f64_lm", iters=4, **okw):
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from tests.test_gpu_lm import build_objective
    g = load_golden(name)
    obj, poses = build_objective(th, g, device="cpu")
    opt = th.LevenbergMarquardt(obj, linearization_kwargs=dict(kernels=OracleKernels()), max_iterations=iters,
                                abs_err_tolerance=0.0, rel_err_tolerance=0.0, **okw)
    return th, g, obj, opt, th.TheseusLayer(opt)


def test_in_place_edit_of_an_auxiliary_tensor_is_seen():
    """A cost weight held by a Variable and scaled IN PLACE (what a torch optimizer does to an nn.Parameter) must change the
    next forward(): the reference re-reads var.tensor at every evaluation (core/objective.py:813-830)."""
    th, g, obj, opt, layer = _layer(iters=1)
    w_vars = [c.weight.diagonal for c in obj.cost_functions.values() if hasattr(c.weight, "diagonal")]
    start = {k: v.tensor.clone() for k, v in obj.optim_vars.items()}
    _, info1 = layer.forward(None, optimizer_kwargs=dict(track_err_history=True, damping=1e-3))
    with torch.no_grad():
        for w in w_vars:
            w.tensor.mul_(3.0)          # no Variable.update(): only the tensor's version counter moves
    _, info2 = layer.forward(start, optimizer_kwargs=dict(track_err_history=True, damping=1e-3))
    e1, e2 = info1.err_history[:, 0], info2.err_history[:, 0]
    assert (e2 > 5.0 * e1).all(), (e1, e2)   # Between errors carry w^2 = 9x; the priors keep theirs


def test_swapping_the_storage_under_an_auxiliary_tensor_is_seen():
    """``var.tensor.data = other`` / ``tensor.set_()`` keep the tensor OBJECT and its version counter and move only the storage
    (the auxiliary stamp keyed on the object alone missed it; key = object + storage + version)."""
    th, g, obj, opt, layer = _layer(iters=1)
    w_vars = [c.weight.diagonal for c in obj.cost_functions.values() if hasattr(c.weight, "diagonal")]
    start = {k: v.tensor.clone() for k, v in obj.optim_vars.items()}
    _, info1 = layer.forward(None, optimizer_kwargs=dict(track_err_history=True, damping=1e-3))
    for w in w_vars:
        t = w.tensor
        ver = t._version
        t.data = t.detach() * 3.0
        assert t._version == ver and w.tensor is t
    _, info2 = layer.forward(start, optimizer_kwargs=dict(track_err_history=True, damping=1e-3))
    e1, e2 = info1.err_history[:, 0], info2.err_history[:, 0]
    assert (e2 > 5.0 * e1).all(), (e1, e2)


def test_variables_view_the_packed_state_outside_unrolled_differentiation():
    """Pose tensors that require grad + a plain (no_grad / implicit) optimize(): the variables are re-pointed at the packed state
    as everywhere else -- only an optimize() that differentiates from the initial tensors (UNROLL / TRUNCATED) leaves them on the
    caller's graph-carrying tensors."""
    th, g, obj, opt, layer = _layer(iters=2)
    for v in obj.optim_vars.values():
        v.update(v.tensor.clone().requires_grad_(True))
    seen = []

    def cb(optimizer, info, delta, it):
        packed = optimizer.linear_solver.linearization.packed
        v = next(iter(obj.optim_vars.values()))
        seen.append(v.tensor.data_ptr() == packed.state[0].data_ptr() or not v.tensor.requires_grad)
    with torch.no_grad():
        opt.optimize(damping=1e-3, end_iter_callback=cb)
    assert seen and all(seen)
    assert not opt.linear_solver.linearization.packed._keep_graph_tensors


def test_in_place_edit_without_any_update_is_seen():
    th, g, obj, opt, layer = _layer(iters=1)
    layer.forward(None, optimizer_kwargs=dict(damping=1e-3))
    packed = opt.linear_solver.linearization.packed
    e0 = packed.error_metric().clone()
    with torch.no_grad():
        for c in obj.cost_functions.values():
            if hasattr(c.weight, "diagonal"):
                c.weight.diagonal.tensor.mul_(2.0)
    info = opt.optimize(track_err_history=True, damping=1e-3)   # nobody called update(): the deep stamp must catch it
    assert (info.err_history[:, 0] > 2.0 * e0).all()


def test_a_second_forward_does_not_overwrite_the_first_solution():
    th, g, obj, opt, layer = _layer(iters=4)
    sol1, _ = layer.forward(None, optimizer_kwargs=dict(damping=1e-3))
    keep = {k: v.clone() for k, v in sol1.items()}
    sol2, _ = layer.forward(None, optimizer_kwargs=dict(damping=1e-3))   # continues from sol1, no update() in between
    for k in keep:
        assert torch.equal(sol1[k], keep[k]), k                           # sol1's tensors were not used as scratch
    assert any(not torch.equal(sol2[k], keep[k]) for k in keep)


def test_exception_inside_optimize_does_not_poison_the_next_forward():
    """an exception after privatize_state() used to leave ``_vars_stale`` set; the next forward(new_inputs) then
    flushed the stale private buffer over the user's new tensors and optimised the OLD problem."""
    th, g, obj, opt, layer = _layer(iters=3)
    start = {k: v.tensor.clone() for k, v in obj.optim_vars.items()}
    layer.forward(None, optimizer_kwargs=dict(damping=1e-3))          # state buffer handed out -> next optimize privatizes
    boom = RuntimeError("kernel error stand-in")

    def raising_complete_step(*a, **k):
        raise boom
    orig = opt._complete_step
    opt._complete_step = raising_complete_step
    with pytest.raises(RuntimeError):
        layer.forward(None, optimizer_kwargs=dict(damping=1e-3))
    opt._complete_step = orig
    packed = opt.linear_solver.linearization.packed
    assert not packed._vars_stale
    # new inputs: the start poses again -- the first error of the history must be the START problem's, as a fresh optimizer sees it
    _, info = layer.forward(start, optimizer_kwargs=dict(track_err_history=True, damping=1e-3))
    th2, g2, obj2, opt2, layer2 = _layer(iters=3)
    _, fresh = layer2.forward(None, optimizer_kwargs=dict(track_err_history=True, damping=1e-3))
    assert torch.allclose(info.err_history, fresh.err_history, rtol=1e-12, atol=0)


def test_best_iter_and_state_history():
    th, g, obj, opt, layer = _layer("pg_f64_lm_adaptive_rejects", iters=6)
    _, kw_ = None, None
    _, _, kw = golden_problem(g)
    kw = {k: v for k, v in kw.items() if k not in ("gauss_newton", "max_iterations", "step_size")}
    _, info = layer.forward(None, optimizer_kwargs=dict(track_best_solution=True, track_err_history=True, **kw))
    h = info.err_history[:, :info.iters_done + 1]
    # nonlinear_optimizer.py:200-202: best_iter = index of the iteration whose error first reached the minimum (0 if never improved)
    for b in range(h.shape[0]):
        best, bi = h[b, 0].item(), 0
        for it in range(info.iters_done):
            if h[b, it + 1].item() < best:
                best, bi = h[b, it + 1].item(), it
        assert int(info.best_iter[b]) == bi
        assert float(info.best_err[b]) == pytest.approx(best)


@pytest.mark.parametrize("name", ["pg_f64_lm", "pg_f64_lm_adaptive_rejects", "pg_f64_lm_converges"])
def test_state_history_matches_the_iterates(name):
    """track_state_history (nonlinear_optimizer.py:150-163,174-178): name -> (B, 3, 4, max_iterations + 1), slot 0 the initial
    value, slot k the variable after counted iteration k, unreached slots inf -- recorded on the device by the sync-free loop
    (device-side slot index: all-rejected attempts are not counted) and by the synchronous loop alike; checked against the
    iterates an end_iter_callback sees."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from tests.test_gpu_lm import build_objective
    g = load_golden(name)
    _, _, kw = golden_problem(g)
    kw = {k: v for k, v in kw.items() if k not in ("gauss_newton", "max_iterations", "step_size")}
    tol = dict(abs_err_tolerance=1e-10, rel_err_tolerance=1e-4) if "converges" in name else dict(abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    out = {}
    for lazy in (True, False):
        obj, _ = build_objective(th, g, device="cpu")
        opt = th.LevenbergMarquardt(obj, linearization_kwargs=dict(kernels=OracleKernels()), max_iterations=8, **tol)
        seen = []
        okw = dict(track_state_history=True, track_err_history=True, **kw)
        if not lazy:
            okw["end_iter_callback"] = lambda o, i, d, it: seen.append({k: v.tensor.clone() for k, v in o.objective.optim_vars.items()})
        start = {k: v.tensor.clone() for k, v in obj.optim_vars.items()}
        _, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=okw)
        out[lazy] = (info, seen, start)
    (ia, _, start), (ib, seen, _) = out[True], out[False]
    K = 8
    for k_, h in ia.state_history.items():
        assert h.shape == start[k_].shape + (K + 1,) and h.dtype == torch.get_default_dtype()
        assert torch.equal(h, ib.state_history[k_])                                    # sync-free == synchronous
        np.testing.assert_allclose(h[..., 0].numpy(), start[k_].numpy(), rtol=0, atol=1e-6)
        for it, st in enumerate(seen):                                                  # what the callback saw after iteration it
            np.testing.assert_allclose(h[..., it + 1].numpy(), st[k_].numpy(), rtol=0, atol=1e-6)
        last = ia.iters_done + (1 if "converges" in name else 0)    # (the converging iteration is recorded, not counted: :202-203)
        assert torch.isfinite(h[..., :last + 1]).all() and torch.isinf(h[..., last + 1:]).all()
    assert ia.iters_done == ib.iters_done


def test_global_params_mirror(monkeypatch):
    """theseus_amd.set_global_params takes the reference's option names (theseus/global_params.py:46-80,
    torchlie/global_params.py:44-68) for the thresholds this path reads; fast_approx_local_jacobians is refused loudly."""
    import theseus_amd as th
    import theseus_amd.kernels as tk
    monkeypatch.setattr(tk, "_REFERENCE_PARAMS", None)   # (stand-alone use: theseus_amd.plugin, if imported, reads the reference's)
    try:
        th.set_global_params({"so3_near_zero_eps_float32": 0.5, "so3_d_near_zero_eps_float64": 0.25, "se2_d_near_zero_eps_float32": 0.75})
        assert tk.lie_eps(torch.float32).near_zero == 0.5 and tk.lie_eps(torch.float64).d_near_zero == 0.25
        assert tk.se2_eps(torch.float32).d_near_zero == 0.75 and tk.lie_eps(torch.float32).near_pi == 1e-2
        with pytest.raises(ValueError):
            th.set_global_params({"so3_quat_eps_float32": 1.0})     # a reference option this path never reads
        th.set_global_params({"fast_approx_local_jacobians": True})
        with pytest.raises(NotImplementedError, match="fast_approx_local_jacobians"):
            _layer()
    finally:
        th.reset_global_params()
    assert tk.lie_eps(torch.float32).near_zero == 1e-2 and not tk.fast_approx_local_jacobians()
    _layer()


def test_check_singular_zeroes_the_singular_items():
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from tests.test_gpu_lm import build_objective
    g = dict(load_golden("pg_f64_lm_adaptive_ellips"))     # batched DiagonalCostWeights
    B = g["poses0"].shape[0]
    g["w_between"] = g["w_between"].copy()
    g["w_prior"] = np.repeat(g["w_prior"], B, axis=0).copy()
    g["w_between"][2] = 0.0
    g["w_prior"][2] = 0.0
    obj, _ = build_objective(th, g, device="cpu")
    solver = th.HipCholeskySolver(obj, linearization_kwargs=dict(kernels=OracleKernels()), check_singular=True)
    obj.update()
    solver.linearization.linearize()
    with pytest.warns(RuntimeWarning, match="Singular matrix found in batch"):
        delta = solver.solve(damping=0.1, ellipsoidal_damping=False)
    assert (delta[2] == 0).all() and all((delta[b] != 0).any() for b in (0, 1, 3))
    plain = th.HipCholeskySolver(obj, linearization_kwargs=dict(kernels=OracleKernels()))
    plain.linearization.linearize()
    np.testing.assert_allclose(plain.solve(damping=0.1, ellipsoidal_damping=False)[[0, 1, 3]].numpy(), delta[[0, 1, 3]].numpy())


def test_update_skips_the_variable_walk_only_when_batch_and_device_are_unchanged():
    """Objective.update(input_tensors) re-resolves the batch size (a pass over every variable) unless every new tensor keeps its
    variable's batch size and device and nothing else was edited since the last resolve (theseus/core/objective.py:708-811)."""
    import theseus_amd as th
    dt = torch.float64
    obj = th.Objective(dtype=dt)
    a = th.Vector(tensor=torch.zeros(2, 3, dtype=dt), name="a")
    b = th.Vector(tensor=torch.zeros(1, 3, dtype=dt), name="b")
    obj.add(th.Difference(a, b, th.ScaleCostWeight(torch.ones(1, dtype=dt)), name="d"))
    calls = {"n": 0}
    walk = obj._resolve_batch_size

    def counted():
        calls["n"] += 1
        return walk()
    obj._resolve_batch_size = counted
    obj.update({"a": torch.ones(2, 3, dtype=dt)})
    assert obj.batch_size == 2 and calls["n"] == 1                      # first update: resolved
    obj.update({"a": torch.full((2, 3), 2.0, dtype=dt)})
    assert obj.batch_size == 2 and calls["n"] == 1                      # same shapes: the walk is skipped
    assert float(a.tensor[0, 0]) == 2.0
    obj.update({"a": torch.ones(5, 3, dtype=dt)})
    assert obj.batch_size == 5 and calls["n"] == 2                      # another batch size: resolved again
    b.update(torch.ones(1, 3, dtype=dt))                                 # an edit outside update(): the next update() walks again
    obj.update({"a": torch.zeros(5, 3, dtype=dt)})
    assert obj.batch_size == 5 and calls["n"] == 3
    with pytest.raises(ValueError):
        obj.update({"a": torch.zeros(5, 3, dtype=dt), "b": torch.zeros(4, 3, dtype=dt)})   # 5 vs 4: not broadcastable


def test_one_repointing_per_forward_and_the_inputs_are_never_written():
    """The optimizer's first sync packs the caller's tensors into a PRIVATE state buffer and leaves the variables on the tensors they
    hold; they are re-pointed ONCE, to views of the final state, when the loop is done (PackedPoseGraph.sync, ``_defer_repoint``) --
    re-pointing them to the packed copy first was 5 ms of host time per forward at 4096 poses.  The caller's tensors keep their
    values; an end_iter_callback still sees the current iterate (the loop flushes before anybody can look)."""
    th, g, obj, opt, layer = _layer(iters=3)
    packed = opt.linear_solver.linearization.packed
    start = {k: v.tensor.clone() for k, v in obj.optim_vars.items()}
    layer.forward(None, optimizer_kwargs=dict(damping=1e-3))           # (first call: packs, builds the structure)
    calls = {"n": 0}
    orig = packed._repoint_variables

    def counted():
        calls["n"] += 1
        return orig()
    packed._repoint_variables = counted
    inputs = {k: v.clone() for k, v in start.items()}
    sol, _ = layer.forward(inputs, optimizer_kwargs=dict(damping=1e-3))
    assert calls["n"] == 1
    for k, v in inputs.items():
        assert torch.equal(v, start[k])                                  # never written into
        assert not torch.equal(sol[k], start[k])                         # (the loop moved)
        assert sol[k] is obj.optim_vars[k].tensor
    assert packed.buffer_of([v.tensor for v in packed.pose_vars]) is not None   # the variables view ONE buffer: the final state
    # with a callback: the variables show the iterate of each iteration (one re-pointing per look)
    seen = []
    calls["n"] = 0
    sol2, _ = layer.forward({k: v.clone() for k, v in start.items()}, optimizer_kwargs=dict(
        damping=1e-3, end_iter_callback=lambda o, i, d, it: seen.append({k: v.tensor.clone() for k, v in o.objective.optim_vars.items()})))
    assert len(seen) == 3 and calls["n"] >= 3
    for k in sol:
        assert torch.equal(sol2[k], sol[k]) and torch.equal(seen[-1][k], sol[k])
        assert not torch.equal(seen[0][k], seen[-1][k])

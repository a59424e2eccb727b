"""Host control logic: replay the scalar trace of a full reference run
(tests/golden/trace_*.npz) through the product's per-image state machine and sampling code
and demand EXACT agreement with what the reference decided at every step."""
import numpy as np

from dorpatch_amd.attack import _ImageState, draw_indices


def _failed(trace, k):
    o = trace["failed_offsets"]
    return trace["failed_values"][o[k]:o[k + 1]].tolist()


def test_state_machine_replays_reference(golden_trace):
    t = golden_trace
    S = int(t["S"])
    n = len(t["i"])
    stages = t["stage"]
    checked = 0
    for stage in (0, 1):
        rows = np.nonzero(stages == stage)[0]
        st = _ImageState(float(t["lr0"]) if "lr0" in t.files else 1e-2,
                         t["structured"][rows[0]], t["coeff_group_lasso"][rows[0]], True, int(t["y0"][0]))
        st.failed_idxs = _failed(t, rows[0])
        for k in rows:
            i = int(t["i"][k])
            # state the reference held when it sampled for step i
            assert np.float32(st.lr_current) == t["lr"][k], (stage, i)
            assert st.structured == t["structured"][k], (stage, i)
            assert st.coeff_group_lasso == t["coeff_group_lasso"][k], (stage, i)
            if i % 100 == 0:            # collect_failure refreshed the list (attack.py:187-190)
                st.failed_idxs = _failed(t, k)
            assert list(st.failed_idxs) == _failed(t, k), (stage, i)
            assert st.not_decay == t["not_decay"][k]
            assert st.n_from_failure(i, S, 1000) == t["n_form_failure"][k]
            if not t["complete"][k]:
                continue                # last recorded step of the stage: results not observable
            save, stop = st.step(i, stage, t["loss_adv"][k], t["loss_target"][k], t["idx"][k],
                                 int(t["n_form_failure"][k]))
            assert save == bool(t["save_best"][k]), (stage, i)
            assert not stop
            checked += 1
    assert checked > 800


def test_sampling_reproduces_reference_rng(golden_trace):
    """np.random.seed(1234) + the product's draw_indices == the reference's sampled indices,
    including the failure-biased phase (i >= 1000)."""
    t = golden_trace
    S = int(t["S"])
    np.random.seed(1234)
    choices = np.arange(2520)
    for k in range(len(t["i"])):
        n_fail = int(t["n_form_failure"][k])
        idx = draw_indices(np.random, _failed(t, k), n_fail, S, choices)
        assert np.array_equal(idx, t["idx"][k]), (int(t["stage"][k]), int(t["i"][k]))


def test_untargeted_run_and_its_switch_replay_the_reference(golden_trace_untargeted):
    """An UNTARGETED reference run (y = None) through the untargeted -> targeted switch at iteration 500 of stage 0
    (attack.py:169-182): the product's per-image state machine, its ``_set_target`` vote over the recorded predictions
    and its sampling code must reproduce every decision of the unmodified reference — label, reset of lr / loss_best /
    not_decay / num_failure, re-collected failure list, and everything test_state_machine_replays_reference checks."""
    from dorpatch_amd.attack import HotLoop

    class Stub(object):          # HotLoop._set_target only touches self.img
        pass

    t = golden_trace_untargeted
    S, stages = int(t["S"]), t["stage"]
    assert not t["targeted"][0] and t["targeted"][-1] and int(t["y"][0]) == int(t["y0"][0])
    np.random.seed(1234)
    choices = np.arange(2520)
    checked = switched = 0
    st = None
    for stage in (0, 1):
        rows = np.nonzero(stages == stage)[0]
        if stage == 0:
            st = _ImageState(float(t["lr0"]), t["structured"][rows[0]], t["coeff_group_lasso"][rows[0]], False, int(t["y0"][0]))
        else:                     # attack.py:124-132: a new stage resets the schedule, not the label / criterion
            st.reset_stage()
            st.structured, st.coeff_group_lasso = float(t["structured"][rows[0]]), float(t["coeff_group_lasso"][rows[0]])
        st.failed_idxs = _failed(t, rows[0])
        for k in rows:
            i = int(t["i"][k])
            if stage == 0 and i == 500 and not st.flag_targeted:       # HotLoop.step, attack.py:169-182
                stub = Stub()
                stub.img = [st]
                st.flag_targeted = True
                changed = HotLoop._set_target(stub, 0, t["pred"][k - 1])
                st.reset_stage()
                st.failed_idxs = _failed(t, k)                         # collect_failure at the switch
                assert changed and st.crit_targeted
                switched += 1
            assert st.y == int(t["y"][k]) and st.flag_targeted == bool(t["targeted"][k]), (stage, i)
            assert np.float32(st.lr_current) == t["lr"][k], (stage, i)
            assert st.structured == t["structured"][k] and st.coeff_group_lasso == t["coeff_group_lasso"][k], (stage, i)
            if i % 100 == 0:
                st.failed_idxs = _failed(t, k)
            assert list(st.failed_idxs) == _failed(t, k), (stage, i)
            assert st.not_decay == t["not_decay"][k], (stage, i)
            n_fail = st.n_from_failure(i, S, 1000)
            assert n_fail == t["n_form_failure"][k]
            assert np.array_equal(draw_indices(np.random, st.failed_idxs, n_fail, S, choices), t["idx"][k]), (stage, i)
            if not t["complete"][k]:
                continue
            save, stop = st.step(i, stage, t["loss_adv"][k], t["loss_target"][k], t["idx"][k], n_fail)
            assert save == bool(t["save_best"][k]), (stage, i)
            assert not stop
            checked += 1
    assert switched == 1 and checked > 1000


def test_lr_floor_and_stop_rule():
    """lr 0.01 -> 0.001 is NOT < 1e-3 in fp32; a second decay is needed."""
    st = _ImageState(1e-2, 1e-3, 1e-5, True, 0)
    st.failed_idxs = []
    loss = np.full(4, 0.5, np.float32)      # never successful, never improving after the first save
    stops = []
    for i in range(500):
        _, stop = st.step(i, 1, loss, 1.0, np.arange(4), 0)
        stops.append(stop)
        if stop:
            break
    assert stops.index(True) == 402           # 1 improving step + 2 x 201 patience windows
    assert st.lr_current < np.float32(1e-3)


def test_sweep_plan_presents_one_row_count():
    """attack.sweep_plan: every group of the failure sweep is g images x rows // g masks with g a power of two that DIVIDES
    ``rows`` — the forward's row count is ``rows`` for every group, whatever the number of running images."""
    from dorpatch_amd.attack import sweep_plan
    for B, rows in ((64, 512), (3, 512), (5, 100), (7, 96), (1, 512), (4, 128), (9, 1)):
        plan = sweep_plan(B, rows)
        assert plan[0][0] == 0 and plan[-1][1] == B and all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
        for b0, b1, masks in plan:
            g = b1 - b0
            assert g & (g - 1) == 0 and g * masks == rows, (B, rows, plan)

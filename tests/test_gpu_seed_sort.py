"""GPU parity of the exact LSD seed order (closes definition D1): the kernel's replay of libstdc++'s std::__introsort_loop
(csrc/seed_sort_kernels.hip) against the library itself, and the line front-end in PLP_SEED_ORDER_LIBSTDCXX mode against an oracle that
sorts its seeds with std::sort exactly as OpenCV's lsd.cpp does (reached from the reference's LSDDetector_custom.cpp:244-257)."""
import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
from plp import plp, synth
from test_index_models import _seed_entries

pytestmark = pytest.mark.gpu

EP = ("startPointX", "startPointY", "endPointX", "endPointY")


def _want(e, depth):
    w = O.std_introsort_loop_entries(e, depth)
    return plp.model_seed_introsort(e, depth) if w is None else w


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("n", [17, 18, 33, 48, 49, 64, 65, 66, 127, 128, 129, 130, 256, 257, 512, 513, 1000, 4095, 4096, 4097, 4098, 9000, 24575, 24576, 24577, 24578, 30000, 76241, 115753, 229401])
def test_kernel_introsort_loop_equals_libstdcxx_at_the_size_thresholds(n, variant):
    """every boundary between the kernel's regimes (lane / wave / workgroup in LDS / workgroup in global memory), on eight key distributions, in both
    configurations the library launches (variant 0: 4 waves and a 4096-entry window, large batches; 1: 16 waves and 24576 entries, small ones)"""
    r = np.random.default_rng(n)
    for kind in range(8):
        e = _seed_entries(r, n, kind)
        assert np.array_equal(plp.seed_introsort_debug(e, variant=variant), _want(e, -1)), (n, kind)


def test_kernel_introsort_loop_random_sizes_and_forced_recursion_budgets():
    """budgets 0..8 reach the heap sort in every regime (a real image never does: the library's budget is 2 * floor(log2 n))"""
    r = np.random.default_rng(5)
    for trial in range(160):
        n = int(r.integers(17, 600)) if trial % 4 == 0 else int(r.integers(600, 9000)) if trial % 4 == 1 else int(r.integers(9000, 40000)) if trial % 4 == 2 else int(r.integers(40000, 120000))
        e = _seed_entries(r, n, trial % 8)
        depth = -1 if trial % 3 else int(r.integers(0, 9))
        if depth >= 0 and n > 30000:
            n = 30000 + trial; e = e[:n]                      # the heap sort of a long segment is one lane's work
        assert np.array_equal(plp.seed_introsort_debug(e, depth, variant=trial % 2), _want(e, depth)), (trial, n, depth)


def test_kernel_leaves_parts_below_the_skip_key_alone_exactly_as_the_model_does():
    """with a skip key (the bin of the smallest defined gradient magnitude in a frame) the kernel and the host model make the same choices:
    identical arrays, and the order of the entries at or above the key is std::sort's"""
    r = np.random.default_rng(9)
    key = lambda a: (a >> np.uint32(20)).astype(np.int64)
    n_dead_total = 0
    for trial in range(64):
        n = [300, 5000, 24577, 40000, 76241, 100000][trial % 6] + trial
        e = _seed_entries(r, n, [1, 7, 0, 2, 3, 6, 4, 5][trial % 8])
        skip = int(r.integers(1, 60)) if trial % 2 else int(r.integers(1, 1024))
        got, n_live = plp.seed_introsort_debug(e, -1, skip, variant=(trial // 2) % 2, return_live=True)
        want = plp.model_seed_introsort(e, -1, skip)
        # round 6: behind n_live lie right parts of global-memory partitions whose pivot key was below the skip key -- the kernel no longer stores into them (nothing
        # reads them again): the live part must be the model's, and the model's entries behind it must all be below the skip key
        assert 0 < n_live <= n and np.array_equal(got[:n_live], want[:n_live]), (trial, n, skip, n_live)
        assert np.all(key(want[n_live:]) < skip), (trial, n, skip, n_live)
        fin = got[:n_live][np.argsort(-key(got[:n_live]), kind="stable")]
        ref = O.std_sort_entries(e)
        assert np.array_equal(fin[key(fin) >= skip], ref[key(ref) >= skip]), (trial, n, skip)
        n_dead_total += n - n_live
    assert n_dead_total > 0, "no trial exercised the dead-suffix rule"


def test_exact_seed_order_leaves_no_key_line_different_from_std_sort(golden_dir):
    """tests/test_d1_seed_order_cost.py measures 3.5 % of the key lines differing between the stable order and std::sort; in the exact mode
    the library must give the std::sort result on every frame of that measurement: 0 differing key lines, raw segments and LBD rows included"""
    frames = [np.asarray(Image.open(golden_dir / n)) for n in ("equirect1_640x480.png", "equirect2_640x480.png", "equirect1_crop_640x480.png", "equirect2_crop_640x480.png")]
    frames += [synth.canvas(1234, 480, 640)] + list(synth.replay(1234, 64, 480, 640))
    lt = plp.LineFeatureTracker()
    lt.set_seed_order(plp.SEED_ORDER_LIBSTDCXX)
    n_lines = n_stable_diff = 0
    for i, f in enumerate(frames):
        ora = O.LineOracle(f, stable_order=False)
        lt.set_grow_waves(1 if i % 2 else 0)
        kl, lbd, fn = lt.extract_LSD_LBD(f)
        assert np.array_equal(lt.debug_read(lt.DBG_RAW), ora.raw), i
        assert np.array_equal(kl, ora.keylsd) and np.array_equal(lbd, ora.lbd) and np.array_equal(fn, ora.linefn), i
        n_lines += len(kl)
        st = O.LineOracle(f, stable_order=True).keylsd
        a = np.stack([kl[k] for k in EP], 1); b = np.stack([st[k] for k in EP], 1)
        n_stable_diff += sum(1 for row in a if not (len(b) and (np.abs(b - row).max(1) == 0).any()))
    assert n_lines > 2500
    assert n_stable_diff > 0, "the two orders are known to differ on these frames: the exact mode must not have fallen back to the stable one"


def test_exact_seed_order_in_a_large_batch_takes_the_other_kernel_configuration_with_the_same_result():
    """batches above 256 frames run the 4-wave configuration of the sort, smaller ones the 16-wave one: 288 frames (24 distinct) against the oracle"""
    import torch
    uniq = synth.replay(78, 24)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(uniq).to(dev).repeat(12, 1, 1).contiguous()
    B, cap = d.shape[0], 512
    d_kl = torch.zeros((B, cap, 68), dtype=torch.uint8, device=dev); d_lbd = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_fn = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev); d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    lt = plp.LineFeatureTracker()
    lt.extract_batch(d, d_kl, d_lbd, d_fn, d_cnt)
    torch.cuda.synchronize()
    lt.last_batch_status()
    cnt = d_cnt.cpu().numpy()
    kl = d_kl.cpu().numpy().view(plp.KL_DTYPE).reshape(B, cap)
    lbd = d_lbd.cpu().numpy()
    for f in range(24):
        ora = O.LineOracle(uniq[f], stable_order=False)
        for b in (f, f + 24 * 5, f + 24 * 11):
            assert cnt[b] == len(ora.keylsd) and np.array_equal(kl[b, :cnt[b]], ora.keylsd) and np.array_equal(lbd[b, :cnt[b]], ora.lbd), (f, b)


@pytest.mark.parametrize("shape", [(480, 752), (376, 1241), (240, 320)])     # EuRoC, KITTI (its first partitions' chunk masks do not fit LDS: the HBM form), a quarter frame (one window after one partition)
def test_large_batches_of_other_geometries_in_the_exact_order(shape):
    """the 4-wave configuration (batches above 256 frames) at the other BASELINE geometries: 264 frames (12 distinct) against the oracle's std::sort order"""
    import torch
    uniq = synth.replay(5 + shape[0], 12, shape[0], shape[1])
    dev = torch.device("cuda:0")
    d = torch.from_numpy(uniq).to(dev).repeat(22, 1, 1).contiguous()
    B, cap = d.shape[0], 512
    assert B > 256
    d_kl = torch.zeros((B, cap, 68), dtype=torch.uint8, device=dev); d_lbd = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_fn = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev); d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    lt = plp.LineFeatureTracker()
    lt.extract_batch(d, d_kl, d_lbd, d_fn, d_cnt)
    torch.cuda.synchronize()
    lt.last_batch_status()
    cnt = d_cnt.cpu().numpy()
    kl = d_kl.cpu().numpy().view(plp.KL_DTYPE).reshape(B, cap)
    lbd = d_lbd.cpu().numpy()
    for f in range(12):
        ora = O.LineOracle(uniq[f], stable_order=False)
        for b in (f, f + 12 * 9, f + 12 * 21):
            assert cnt[b] == len(ora.keylsd) and np.array_equal(kl[b, :cnt[b]], ora.keylsd) and np.array_equal(lbd[b, :cnt[b]], ora.lbd), (shape, f, b)
        want_order = np.asarray(ora.order)
        s_ = ora.scaled.astype(np.int64)
        DA = s_[1:, 1:] - s_[:-1, :-1]; BC = s_[:-1, 1:] - s_[1:, :-1]
        dmask = np.zeros(ora.scaled.shape, bool)
        dmask[:-1, :-1] = ~(np.sqrt(((DA + BC) ** 2 + (DA - BC) ** 2) / 4.0) <= 2.0 / np.sin(np.pi * 22.5 / 180))
        assert np.array_equal(lt.debug_read(lt.DBG_ORDER, f + 12 * 9), want_order[dmask.ravel()[want_order]]), (shape, f)


def test_exact_seed_order_in_a_batch_and_against_the_single_frame_path():
    import torch
    frames = synth.replay(77, 6)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(frames).to(dev)
    B, cap = len(frames), 512
    d_kl = torch.zeros((B, cap, 68), dtype=torch.uint8, device=dev)
    d_lbd = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_fn = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    lt = plp.LineFeatureTracker()
    lt.set_seed_order(plp.SEED_ORDER_LIBSTDCXX)
    lt.extract_batch(d, d_kl, d_lbd, d_fn, d_cnt)
    torch.cuda.synchronize()
    lt.last_batch_status()
    cnt = d_cnt.cpu().numpy()
    kl = d_kl.cpu().numpy().view(plp.KL_DTYPE).reshape(B, cap)
    for f in range(B):
        ora = O.LineOracle(frames[f], stable_order=False)
        assert cnt[f] == len(ora.keylsd)
        assert np.array_equal(kl[f, :cnt[f]], ora.keylsd)
        assert np.array_equal(d_lbd[f, :cnt[f]].cpu().numpy(), ora.lbd)
    # switching the mode of a live context back and forth
    lt.set_seed_order(plp.SEED_ORDER_STABLE)
    k2, _, _ = lt.extract_LSD_LBD(frames[0])
    assert np.array_equal(k2, O.LineOracle(frames[0], stable_order=True).keylsd)
    lt.set_seed_order(plp.SEED_ORDER_LIBSTDCXX)
    k3, _, _ = lt.extract_LSD_LBD(frames[0])
    assert np.array_equal(k3, O.LineOracle(frames[0], stable_order=False).keylsd)

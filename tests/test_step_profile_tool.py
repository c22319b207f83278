"""tools/step_profile.py books every kernel with ITS OWN register allocation (until late in round 6 it matched names by substring and took the maximum: k_lsd_grow was booked with
k_lsd_grow_mw's 170 registers, which put the step's register packing at 0.77 instead of 0.66)."""
import importlib.util
import os

import pytest

spec = importlib.util.spec_from_file_location("step_profile", os.path.join(os.path.dirname(__file__), "..", "tools", "step_profile.py"))
sp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sp)

V = {
    "_ZN3plp10k_lsd_growENS_10LinePlanesENS_9LsdParamsEiii": 111,
    "_ZN3plp13k_lsd_grow_mwENS_10LinePlanesENS_9LsdParamsENS_8MwLayoutE": 170,
    "_ZN3plp5k_lbdENS_10LinePlanesENS_13LbdWeightsDevE": 51,
    "_ZN3plp15k_lbd_match_1nnEPKhPKiiS1_S3_iNS_8MihRanksEPiS5_": 36,
    "_ZN3plp18k_match_topk_lanesILi1EEEvNS_12MatchProblemE": 67,
    "_ZN3plp18k_match_topk_lanesILi2EEEvNS_12MatchProblemE": 89,
    "_ZN3plp18k_match_topk_cellsENS_12MatchProblemEi": 58,
    "_ZN3plp6ss_thr15k_lsd_seed_sortENS_10LinePlanesENS_9LsdParamsEiPjS3_m": 77,
    "_ZN3plp6ss_lat15k_lsd_seed_sortENS_10LinePlanesENS_9LsdParamsEiPjS3_m": 79,
    "_ZN3plp6ss_thr17k_seed_sort_debugEPjiijS1_mPiS2_": 81,
}


@pytest.mark.parametrize("kernel, want", [
    ("plp::k_lsd_grow", 112), ("plp::k_lsd_grow_mw", 176), ("plp::k_lbd", 56), ("plp::k_lbd_match_1nn", 40),
    ("plp::k_match_topk_lanes<1>", 72), ("plp::k_match_topk_lanes<2>", 96), ("plp::k_match_topk_cells", 64),
    ("plp::ss_thr::k_lsd_seed_sort", 80), ("plp::ss_lat::k_lsd_seed_sort", 80), ("plp::k_not_there", None),
])
def test_a_kernel_is_booked_with_its_own_registers(kernel, want):
    assert sp.vg_of(V, kernel) == want


def test_the_bounds_of_a_stored_table():
    per = {"a": {"waves": 1024, "cycles_per_wave": 2.4e6, "vgprs": 512, "launches_per_step": 1, "register_cycles": 1024 * 2.4e6 * 512, "lds_byte_cycles": 256 * 160 * 1024 * 2.4e6}}
    ob = sp.bounds(per, 2.0)
    assert ob["ideal_ms"] == 1.0 and ob["packing"] == 0.5 and ob["lds"]["ideal_ms"] == 1.0 and ob["shares"] == {"a": 1.0}

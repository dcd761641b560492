"""CPU: tile ownership and the sizing of one rank's render state (hydra_hip_tile_owners / hydra_hip_plan_render_state are host
arithmetic inside libhydra_hip.so -- the same routines trace_pass uses -- so the limits are checked here without a GPU)."""
import numpy as np
import pytest


def test_morton_tile_owners_match_the_numpy_restatement_and_balance(built):
    from hydracore_amd.capi import tile_owners
    from hydracore_amd.multi_gpu import tile_owner_mask, tile_owner_table
    for (w, h, world, tile) in ((1920, 1080, 8, 64), (3840, 2160, 8, 64), (96, 96, 3, 16), (64, 48, 2, 16), (1000, 37, 5, 8), (1920, 1080, 1, 64)):
        own = tile_owners(w, h, world, tile)
        assert (own == tile_owner_table(w, h, world, tile)).all()
        counts = np.bincount(own.ravel(), minlength=world)
        assert counts.max() - counts.min() <= 1                      # round-robin over the Morton order
        total = sum(int(tile_owner_mask(w, h, r, world, tile).sum()) for r in range(world))
        assert total == w * h                                        # disjoint and complete
    # neighbouring tiles of a rank's share are spread over the frame: at 1080p / 8 ranks every rank owns tiles in every quarter
    own = tile_owners(1920, 1080, 8, 64)
    for r in range(8):
        ys, xs = np.nonzero(own == r)
        assert ys.min() < 5 and ys.max() > 11 and xs.min() < 8 and xs.max() > 21


def test_render_state_plan_scales_with_one_over_world_and_checks_its_limits(built):
    from hydracore_amd import HydraError
    from hydracore_amd.capi import plan_render_state
    # BASELINE configs[3]: 3840x2160, 8 ranks, bench.py's default of 64 x 8 samples per pixel in flight
    plans = [plan_render_state(3840, 2160, r, 8, 64, 512) for r in range(8)]
    assert sum(p["owned_pixels"] for p in plans) == 3840 * 2160
    for p in plans:
        assert p["samples_in_flight"] == 512 and p["paths"] == p["owned_pixels"] * 512 < 2 ** 31
        assert p["generator_bytes"] == p["owned_pixels"] * 512 * 8 and p["contrib_bytes"] == p["owned_pixels"] * 512 * 16
        assert p["segments"] * p["segment_capacity"] >= p["paths"]
        assert p["total_bytes"] < 150e9                             # fits one MI355X (288 GB) with the scene and room to spare
    one = plan_render_state(1920, 1080, 0, 1, 64, 64)
    eighth = plan_render_state(1920, 1080, 3, 8, 64, 64)
    assert 0.11 < eighth["total_bytes"] / one["total_bytes"] < 0.14   # per-rank state is ~1/8 of the one-rank state
    assert one["path_state_bytes"] == one["segments"] * one["segment_capacity"] * 228
    assert plan_render_state(1920, 1080, 0, 1, 64, 0)["samples_in_flight"] == 16   # chosen from the resolution
    with pytest.raises(HydraError, match="2\\^31"):
        plan_render_state(3840, 2160, 0, 1, 64, 512)               # one rank cannot hold 4.2 G paths: path slots are ints
    with pytest.raises(HydraError, match="2\\^32"):
        plan_render_state(3840, 2160, 0, 8, 64, 518)               # generator slots stream * w * h + pixel are 32-bit
    with pytest.raises(HydraError):
        plan_render_state(1920, 1080, 8, 8, 64, 16)                # rank out of range

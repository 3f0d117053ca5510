"""bench.py's bookkeeping that needs no GPU: K timed steps always render the configuration's spp (K x P passes x S sample lanes,
`plan_steps`), and `roofline.moved_bytes_frac` prices a pass by the bytes the fused kernels really move (`moved_bytes`)."""
import bench


def test_the_drivers_and_the_default_step_counts_render_2000_spp():
    c3 = bench.CONFIGS["c3"]
    assert bench.plan_steps(c3) == (50, 1, 40)                 # python bench.py
    assert bench.plan_steps(c3, 20) == (20, 2, 50)             # the driver: --steps 20
    for k in (1, 2, 4, 5, 8, 10, 20, 25, 40, 50, 100):
        K, P, S = bench.plan_steps(c3, k)
        assert K == k and K * P * S == 2000 and 16 <= S <= 64, (K, P, S)
    for k in (3, 7, 13, 33):                                   # no divisor: a little more than asked, never less
        K, P, S = bench.plan_steps(c3, k)
        assert 2000 <= K * P * S < 2000 + K * S and 16 <= S <= 64
    assert bench.plan_steps(c3, 20, 40) == (20, 3, 40)         # an explicit S is kept, P rounds up


def test_other_configs_keep_their_sample_lanes():
    assert bench.plan_steps(bench.CONFIGS["c2"]) == (16, 1, 32)
    K, P, S = bench.plan_steps(bench.CONFIGS["c2"], 20)
    assert K * P * S >= 512 and 16 <= S <= 64
    for k in (None, 1, 7, 20, 64):                             # the 4K / 1,024-sphere frame: S stays 16 (S x 8.3 M rays per pass; the library takes < 226 M)
        K, P, S = bench.plan_steps(bench.CONFIGS["c5"], k)
        assert S == 16 and K * P * S >= 128


def test_moved_bytes_per_ray_bounce():
    # every ray survives two bounces and ends in the third: bounce 0 reads the 32-B home record, survivors write 76 B, ended paths 36 B
    live = [1000, 1000, 1000]
    assert bench.moved_bytes(live) == 1000 * (32 + 76) + 1000 * (76 + 76) + 1000 * (76 + 36)
    # half the rays end at every bounce
    live = [1024, 512, 256]
    want = 1024 * 32 + 512 * 76 + 512 * 36 + 512 * 76 + 256 * 76 + 256 * 36 + 256 * 76 + 256 * 36
    assert bench.moved_bytes(live) == want
    # always below SURVEY §8(d)'s 152 B per ray-bounce
    assert bench.moved_bytes(live) < 152 * sum(live)
    assert bench.moved_bytes([]) == 0

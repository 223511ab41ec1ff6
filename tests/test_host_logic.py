"""Host-side launch planning (no GPU): how the attention wrapper covers the (q-block, head) grid with whole CU rounds."""


def test_attention_plan_fills_whole_rounds(pkg):
    plan = pkg.native.attention_plan
    # full clip on one GPU: 72 q-blocks x 32 heads = 2304 workgroups = 9 whole rounds -> one unsplit launch
    assert plan(1, 32, 18432, 18432) == [(0, 18432, 1)]
    # 8-way token bands (gather form): 288 workgroups = 1 round + 32 -> 8 q-blocks unsplit, the last one in 8 key chunks
    assert plan(1, 32, 2304, 18432) == [(0, 2048, 1), (2048, 2304, 8)]
    # 8-way head all-to-all: 4 heads x 72 q-blocks, same 288 workgroups
    assert plan(1, 4, 18432, 18432) == [(0, 16384, 1), (16384, 18432, 8)]
    # 2-way: 1152 workgroups = 4 rounds + 128 -> tail in 2 chunks
    assert plan(1, 16, 18432, 18432) == [(0, 16384, 1), (16384, 18432, 2)]
    # less than one round in total: split everything (flash-decoding style)
    (q0, q1, n), = plan(1, 2, 256, 8192)
    assert (q0, q1) == (0, 256) and n == 8
    # tiny: nothing to gain
    assert plan(1, 2, 128, 128) == [(0, 128, 1)]
    # every plan covers [0, Sq) exactly once, cuts on 256-row boundaries
    for args in [(1, 32, 4608, 18432), (5, 32, 256, 256), (2, 32, 2048, 2048), (1, 8, 18432, 18432), (3, 4, 1000, 5000)]:
        p = plan(*args)
        assert p[0][0] == 0 and p[-1][1] == args[2]
        for a, b in zip(p, p[1:]):
            assert a[1] == b[0] and a[1] % 256 == 0

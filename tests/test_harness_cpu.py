"""Host logic of selfc_amd.harness that needs no GPU: GOP slicing / padding and shard-by-GOP."""
from selfc_amd import harness


def test_gop_slices_pad_with_last_frame():
    s = harness.gop_slices(100)
    assert len(s) == 15 and s[0] == list(range(7)) and s[-1] == [98, 99, 99, 99, 99, 99, 99]
    assert harness.gop_slices(14) == [list(range(7)), list(range(7, 14))]          # no extra GOP when divisible
    assert harness.gop_slices(1) == [[0] * 7]


def test_shard_gops_partitions_without_overlap():
    for world in (1, 2, 4, 8):
        parts = [harness.shard_gops(15, r, world) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(15))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1

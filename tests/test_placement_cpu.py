"""Host placement of the ranks (placement.py): the GPU -> PCI -> NUMA-node map is read from sysfs without touching the HIP
runtime; faked here with a temporary tree (two sockets, four GPUs each, as on an MI355X node)."""
import os

from ziragroundingdino_amd import placement


def _fake_node(tmp_path, visible=None):
    kfd, pci = tmp_path / "kfd", tmp_path / "pci"
    for i in range(2):                                   # two CPU nodes first, as KFD lists them
        d = kfd / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\n")
    for g in range(8):
        d = kfd / str(2 + g)
        d.mkdir(parents=True)
        bus = 0x10 + 0x10 * g
        (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\ndomain 0\nlocation_id %d\n" % (bus << 8))
        dev = pci / ("0000:%02x:00.0" % bus)
        dev.mkdir(parents=True)
        node = g // 4
        (dev / "numa_node").write_text("%d\n" % node)
        (dev / "local_cpulist").write_text("%d-%d,%d-%d\n" % (node * 64, node * 64 + 63, 128 + node * 64, 128 + node * 64 + 63))
    return str(kfd), str(pci)


def test_each_rank_gets_its_own_slice_of_its_gpus_numa_node(tmp_path, monkeypatch):
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    kfd, pci = _fake_node(tmp_path)
    assert len(placement.gpu_pci_addresses(kfd)) == 8
    allowed = set(range(256))
    got = [placement.cores_for_local_rank(r, 8, kfd, pci, allowed)[0] for r in range(8)]
    for r, cores in enumerate(got):
        node = r // 4
        local = set(range(node * 64, node * 64 + 64)) | set(range(128 + node * 64, 128 + node * 64 + 64))
        assert cores <= local and len(cores) == 32            # a quarter of the node's 128 hardware threads
    assert all(not (got[a] & got[b]) for a in range(8) for b in range(a + 1, 8))


def test_visible_devices_remap_and_fallback(tmp_path, monkeypatch):
    kfd, pci = _fake_node(tmp_path)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5,6")
    cores, how = placement.cores_for_local_rank(0, 2, kfd, pci, set(range(256)))
    assert min(cores) >= 64 and "NUMA node" in how             # GPU 5 sits on the second socket
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    cores, how = placement.cores_for_local_rank(1, 4, str(tmp_path / "none"), pci, set(range(16)))
    assert cores == {4, 5, 6, 7} and "even split" in how       # no topology: equal slices of what is visible


def test_pin_this_rank_sets_the_affinity_of_this_process(monkeypatch):
    before = os.sched_getaffinity(0)
    try:
        monkeypatch.setenv("LOCAL_RANK", "1")
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
        cores = placement.pin_this_rank(verbose=False)
        assert cores and os.sched_getaffinity(0) == cores and cores <= before
        monkeypatch.delenv("LOCAL_RANK")
        assert placement.pin_this_rank(verbose=False) is None   # not a multi-rank launch: left alone
    finally:
        os.sched_setaffinity(0, before)

"""Where a rank runs on the host: one process per GPU (``torch.distributed.run``), each pinned to the CPU cores of ITS GPU's NUMA
node before anything touches the GPU -- an MI355X node has two sockets with four GPUs each, the encoder pieces of a step are
launched eagerly (graphs.GraphedTransformer: ~700 launches per step that the host must stay ahead of) and eight unpinned
ranks migrate across sockets and share cores.  Nothing here initialises the HIP runtime: the GPU -> PCI device -> NUMA node
map is read from sysfs (``/sys/class/kfd``, ``/sys/bus/pci``); where that fails the visible cores are split evenly by local
rank.  The reference leaves placement to detectron2's launcher (train_multidatasets.py:573-580), which does not pin."""
import os

_KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"
_PCI_DEVICES = "/sys/bus/pci/devices"


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_pci_addresses(kfd_nodes=_KFD_NODES):
    """PCI addresses (dddd:bb:dd.f) of the GPUs in KFD topology order (= HIP's device order without *_VISIBLE_DEVICES)."""
    out = []
    try:
        names = sorted(os.listdir(kfd_nodes), key=lambda s: int(s) if s.isdigit() else 1 << 30)
    except OSError:
        return out
    for d in names:
        try:
            with open(os.path.join(kfd_nodes, d, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) <= 0:      # a CPU node
            continue
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        out.append("%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    return out


def visible_order(n_gpus):
    """Indices into the KFD order that this process's HIP devices 0, 1, ... stand for."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                return [int(x) for x in v.split(",") if x.strip() != ""]
            except ValueError:                           # (UUID form: no map without the runtime)
                return None
    return list(range(n_gpus))


def cores_for_local_rank(local_rank, local_world, kfd_nodes=_KFD_NODES, pci_devices=_PCI_DEVICES, allowed=None):
    """(cores, how): the cores rank ``local_rank`` of ``local_world`` should run on.  ``how`` says which rule applied."""
    allowed = set(os.sched_getaffinity(0)) if allowed is None else set(allowed)
    gpus = gpu_pci_addresses(kfd_nodes)
    order = visible_order(len(gpus))
    if gpus and order is not None and local_rank < len(order) and order[local_rank] < len(gpus):
        dev = gpus[order[local_rank]]
        try:
            with open(os.path.join(pci_devices, dev, "local_cpulist")) as f:
                cores = _parse_cpulist(f.read()) & allowed
        except OSError:
            cores = set()
        if cores:
            # the ranks whose GPUs share this node take equal slices of its cores, in rank order
            mates = [r for r in range(min(local_world, len(order)))
                     if order[r] < len(gpus) and _same_node(gpus[order[r]], dev, pci_devices)]
            if len(mates) > 1 and len(cores) >= len(mates):
                srt, i = sorted(cores), mates.index(local_rank)
                per = len(srt) // len(mates)
                cores = set(srt[i * per:(i + 1) * per])
            return cores, "NUMA node of GPU %s" % dev
    srt = sorted(allowed)
    per = max(1, len(srt) // max(1, local_world))
    cores = set(srt[(local_rank % max(1, local_world)) * per:][:per]) or allowed
    return cores, "even split of the %d visible cores" % len(srt)


def _same_node(a, b, pci_devices):
    def node(dev):
        try:
            with open(os.path.join(pci_devices, dev, "numa_node")) as f:
                return int(f.read().strip())
        except (OSError, ValueError):
            return None
    na, nb = node(a), node(b)
    return na is not None and na == nb


def pin_this_rank(local_rank=None, local_world=None, verbose=True):
    """Pin the calling process (call before the first GPU call).  No-op outside a multi-rank launch unless told otherwise."""
    if local_rank is None:
        if "LOCAL_RANK" not in os.environ:
            return None
        local_rank = int(os.environ["LOCAL_RANK"])
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if local_world <= 1 and os.environ.get("ZIRA_PIN_SINGLE_RANK", "0") != "1":
        return None
    cores, how = cores_for_local_rank(local_rank, local_world)
    try:
        os.sched_setaffinity(0, cores)
    except OSError as exc:
        if verbose:
            print("[placement] rank %d: could not set the CPU affinity (%s)" % (local_rank, exc), flush=True)
        return None
    if verbose:
        srt = sorted(cores)
        print("[placement] local rank %d of %d: %d cores %d..%d (%s)" % (local_rank, local_world, len(srt), srt[0], srt[-1], how),
              flush=True)
    return cores

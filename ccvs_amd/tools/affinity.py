"""Per-rank CPU placement for one-process-per-GPU runs ("weak-scaling risk is host-side").

Each rank of an N-GPU job runs one scheduler thread, 1 + `chains` launch threads, a noise thread and torch's intra-op pool.
Left to the OS scheduler, the ranks' threads migrate across sockets and share cores; here every rank pins itself, BEFORE its
first GPU call, to its own block of cores -- the cores of the NUMA node its GPU hangs off when `rocm-smi --showtoponuma` can be
parsed, split evenly between the ranks that share that node; contiguous equal blocks of the allowed cores otherwise.
The reference leaves placement to the launcher (tools/engine.py:17-60 sets the device only).
"""
import os
import re
import subprocess


def _parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def numa_cpus():
    """{node: [cpus]} from sysfs; {} when not readable."""
    base = "/sys/devices/system/node"
    nodes = {}
    try:
        for name in os.listdir(base):
            m = re.fullmatch(r"node(\d+)", name)
            if m:
                with open(os.path.join(base, name, "cpulist")) as f:
                    nodes[int(m.group(1))] = _parse_cpulist(f.read())
    except OSError:
        return {}
    return nodes


def parse_showtoponuma(text):
    """{gpu index: numa node} from `rocm-smi --showtoponuma` ('GPU[3]\t\t: (Topology) Numa Node: 1')."""
    out = {}
    for m in re.finditer(r"GPU\[(\d+)\]\s*:\s*\(Topology\)\s*Numa Node:\s*(-?\d+)", text):
        out[int(m.group(1))] = int(m.group(2))
    return out


def gpu_numa_nodes(timeout=20.0):
    """{gpu index: numa node} as rocm-smi reports it, {} if the tool is missing or its output does not parse (a child process:
    this process has not touched the GPU yet)."""
    try:
        res = subprocess.run(["rocm-smi", "--showtoponuma"], capture_output=True, text=True, timeout=timeout)
    except (OSError, subprocess.SubprocessError):
        return {}
    return parse_showtoponuma(res.stdout) if res.returncode == 0 else {}


def rank_cpu_set(local_rank, local_world, allowed, gpu_node=None, node_cpus=None):
    """The cores of rank `local_rank` of `local_world` ranks on this host.
    allowed: cores this process may use (sorted list).  gpu_node: {gpu index: numa node} or None; node_cpus: {node: [cpus]}.
    NUMA-aware when every rank's GPU maps to a known node with allowed cores: the ranks of a node split its cores in rank order;
    otherwise contiguous equal blocks of `allowed`.  Never empty: with fewer cores than ranks the ranks share them round-robin."""
    allowed = sorted(allowed)
    if local_world <= 1 or not allowed:
        return allowed
    if gpu_node and node_cpus and all(gpu_node.get(r, -1) in node_cpus for r in range(local_world)):
        node = gpu_node[local_rank]
        mates = [r for r in range(local_world) if gpu_node[r] == node]
        cores = [c for c in node_cpus[node] if c in set(allowed)]
        if len(cores) >= len(mates):
            per = len(cores) // len(mates)
            k = mates.index(local_rank)
            return cores[k * per:(k + 1) * per]
    if len(allowed) < local_world:
        return [allowed[local_rank % len(allowed)]]
    per = len(allowed) // local_world
    return allowed[local_rank * per:(local_rank + 1) * per]


def pin_rank(local_rank=None, local_world=None, numa=True):
    """Pin this process (call before the first GPU call and before torch starts its thread pools) and return the core list;
    None when there is nothing to do (one rank, or `CCVS_AFFINITY=0`)."""
    if os.environ.get("CCVS_AFFINITY", "1") == "0":
        return None
    local_rank = int(os.environ.get("LOCAL_RANK", 0)) if local_rank is None else local_rank
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", 1))) if local_world is None else local_world
    if local_world <= 1:
        return None
    allowed = sorted(os.sched_getaffinity(0))
    gpu_node = gpu_numa_nodes() if numa else {}
    cores = rank_cpu_set(local_rank, local_world, allowed, gpu_node, numa_cpus() if gpu_node else None)
    os.sched_setaffinity(0, cores)
    return cores

"""Multi-GPU sharding of independent MPC problems: contiguous slices per rank, no data-path
collective during the solve, one all-gather of the optimal trajectories per solve (RCCL on GPUs, gloo in CPU tests).

`TrajectoryGather` is the one collective path of the job: bench.py drives it on RCCL with the HIP solver writing into its
local block, tests/test_distributed_gloo.py drives the same object on gloo."""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous slice [lo, hi) of `total` problems owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_capacity(total, world):
    """Largest slice of shard_range(total, world, .): the per-rank block size of the gather (smaller slices are zero padded)."""
    return (total + world - 1) // world


class TrajectoryGather:
    """One flat block [x | u] per rank and ONE all-gather per solve, asynchronous: the collective of solve i runs on the backend's
    own stream while solve i + 1 is being computed and is waited for (`drain`) before the local block is overwritten again.

    x block: capacity x (nodes + 1) x nx, u block: capacity x nodes x nu (doubles), `capacity` problems per rank (a rank that owns
    fewer leaves the tail zero).  On an 8-GPU MI355X node a 512-problem H1 block is 18 MB and crosses each xGMI link once.

    Stream contract (GPU): every method must be called with the solver's stream as torch's current stream
    (`with torch.cuda.stream(s)`), so that the collective is ordered after the export that fills the block and the next export is
    ordered after `drain()` - torch.distributed orders a collective only against the CURRENT stream."""

    def __init__(self, capacity, nodes, nx, nu, device, group=None, dtype=torch.float64, mode="all"):
        """mode "all": all-gather, every rank ends up with every rank's block (default; 7 x 18 MB received per GPU and solve at
        BASELINE.json configs[2]); mode "root": gather to rank 0 only - what the north-star's "final gather" needs at the least -, the
        other ranks only send their block."""
        if mode not in ("all", "root"):
            raise ValueError("gather mode is 'all' or 'root'")
        self.mode = mode
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.capacity, self.nodes, self.nx, self.nu = int(capacity), int(nodes), int(nx), int(nu)
        self.n_x = self.capacity * (self.nodes + 1) * self.nx
        self.n_u = self.capacity * self.nodes * self.nu
        self.local = torch.zeros(self.n_x + self.n_u, dtype=dtype, device=device)
        holds_all = mode == "all" or self.rank == 0
        self.gathered = (torch.zeros(self.world * (self.n_x + self.n_u), dtype=dtype, device=device) if holds_all else None) \
            if self.world > 1 or dist.is_initialized() else self.local
        self.pending = []

    # views of this rank's block (the solver exports straight into them)
    @property
    def x_local(self):
        return self.local[:self.n_x].view(self.capacity, self.nodes + 1, self.nx)

    @property
    def u_local(self):
        return self.local[self.n_x:].view(self.capacity, self.nodes, self.nu)

    def launch(self):
        """Start the all-gather of the local block (no-op without a process group)."""
        if not dist.is_initialized():
            return
        if self.mode == "all":
            self.pending.append(dist.all_gather_into_tensor(self.gathered, self.local, group=self.group, async_op=True))
        else:
            blk = self.n_x + self.n_u
            parts = [self.gathered[r * blk:(r + 1) * blk] for r in range(self.world)] if self.rank == 0 else None
            # dst is a GLOBAL rank: the first rank of the group (rank 0 of the default group)
            dst = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            self.pending.append(dist.gather(self.local, parts, dst=dst, group=self.group, async_op=True))

    def drain(self):
        """Make the current stream (GPU) / the caller (CPU) wait for every collective in flight: afterwards the local block may be
        overwritten and `gathered` may be read."""
        while self.pending:
            self.pending.pop().wait()

    def block(self, rank):
        """(x, u) views of the gathered block of `rank` (mode "root": on rank 0 only)."""
        if self.gathered is None:
            raise RuntimeError("gather mode 'root': only rank 0 holds the gathered trajectories")
        base = rank * (self.n_x + self.n_u)
        flat = self.gathered[base:base + self.n_x + self.n_u]
        return flat[:self.n_x].view(self.capacity, self.nodes + 1, self.nx), flat[self.n_x:].view(self.capacity, self.nodes, self.nu)

    def assemble(self, total):
        """All `total` problems in global order (drops the padding of short shards): x [total, nodes+1, nx], u [total, nodes, nu]."""
        xs, us = [], []
        for r in range(self.world):
            lo, hi = shard_range(total, self.world, r)
            x, u = self.block(r)
            xs.append(x[:hi - lo]); us.append(u[:hi - lo])
        return torch.cat(xs), torch.cat(us)

    def own_block_consistent(self):
        """The gathered copy of this rank's block equals what it sent (mode "root": checked where the copy exists, rank 0)."""
        if self.gathered is None:
            return True
        base = self.rank * (self.n_x + self.n_u)
        return bool(torch.equal(self.gathered[base:base + self.n_x + self.n_u], self.local))


def reduce_stats(values, group=None, op="sum"):
    """Reduce a small float64 vector of per-rank statistics (cost, SSEs, failure counts; elapsed time with op="max")."""
    t = torch.as_tensor(values, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=group)
        t = t.cpu()
    return t


# ---- which algorithm / protocol RCCL picks for the per-solve gather ------------------------------------------------------
# xGMI is point to point (7 links per GPU): for the 9.5 .. 18 MB block of a solve SURVEY.md section 8(e) argues for direct all-pairs transfers over a ring.
# What RCCL really chooses is only visible in its debug log, so `bench.py --gather-report` switches the log on (before the process group exists - the
# variables are read at init), and rank 0 parses its own file after the run into config.distributed.collective.  NCCL_ALGO / NCCL_PROTO pass through
# `--gather-algo` / `--gather-proto`.
_NCCL_ALGOS = {0: "TREE", 1: "RING", 2: "COLLNET_DIRECT", 3: "COLLNET_CHAIN", 4: "NVLS", 5: "NVLS_TREE", 6: "PAT"}
_NCCL_PROTOS = {0: "LL", 1: "LL128", 2: "SIMPLE"}


def nccl_debug_env(log_path, algo=None, proto=None):
    """Environment for an RCCL process group whose choices are to be reported: returns the dict of variables to set BEFORE init_process_group."""
    env = {"NCCL_DEBUG": "INFO", "NCCL_DEBUG_SUBSYS": "INIT,GRAPH,TUNING,COLL", "NCCL_DEBUG_FILE": log_path}
    if algo:
        env["NCCL_ALGO"] = str(algo)
    if proto:
        env["NCCL_PROTO"] = str(proto)
    return env


def parse_nccl_debug(text):
    """What an NCCL / RCCL INFO log says about the collectives that ran: per collective name the (algorithm, protocol, bytes) choices seen, the
    transports of the channels (P2P/IPC = xGMI peer access, SHM, NET), the number of channels / rings and the library version.  Tolerant of the two
    spellings of the tuning line (`Algo 1 proto 2` of older releases, `Algo RING proto SIMPLE` of newer ones)."""
    import re
    out = {"version": None, "collectives": {}, "transports": {}, "channels": None, "forced": {}}
    m = re.search(r"(?:NCCL|RCCL) version\s*:?\s*([0-9][0-9A-Za-z.+:\-]*)", text)      # "NCCL version 2.21.5+hip6.3" / "RCCL version : 2.26.6-HEAD:64f48b6"
    if m:
        out["version"] = m.group(1)
    for name, nbytes, algo, proto in re.findall(r"(\w+): (\d+) Bytes -> Algo (\w+) proto (\w+)", text):
        a = _NCCL_ALGOS.get(int(algo), algo) if algo.isdigit() else algo.upper()
        p = _NCCL_PROTOS.get(int(proto), proto) if proto.isdigit() else proto.upper()
        entry = out["collectives"].setdefault(name, [])
        rec = {"bytes": int(nbytes), "algo": a, "proto": p}
        if rec not in entry:
            entry.append(rec)
    for via in re.findall(r"Channel \d+(?:/\d+)? : \d+\[[0-9a-fx]+\] -> \d+\[[0-9a-fx]+\] (?:\[\w+\] )?via ([\w/]+)", text):
        out["transports"][via] = out["transports"].get(via, 0) + 1
    m = re.search(r"(\d+) coll channels", text) or re.search(r"Connected all rings.*?(\d+) channels", text)
    if m:
        out["channels"] = int(m.group(1))
    for var in ("NCCL_ALGO", "NCCL_PROTO"):
        m = re.search(var + r" set by environment to (\S+)", text)
        if m:
            out["forced"][var] = m.group(1).rstrip(".")
    return out

"""bench.py --gather-report: what RCCL chose for the per-solve gather is parsed out of its INFO log (bipedal_control_amd.distributed.parse_nccl_debug).
The log excerpts below are written after the format strings of NCCL 2.18 .. 2.21 (numeric and symbolic algorithm names)."""
from bipedal_control_amd import distributed as bd

OLD = """
node:101:101 [0] NCCL INFO NCCL_ALGO set by environment to Ring
node:101:140 [0] NCCL INFO RCCL version 2.18.6+hip6.1 HEAD:abcdef
node:101:140 [0] NCCL INFO Channel 00/0 : 0[c000] -> 1[1c000] via P2P/IPC/read
node:101:140 [0] NCCL INFO Channel 01/0 : 0[c000] -> 1[1c000] via P2P/IPC/read
node:101:140 [0] NCCL INFO Channel 00/0 : 7[e9000] -> 0[c000] via P2P/IPC/read
node:101:140 [0] NCCL INFO Connected all rings
node:101:140 [0] NCCL INFO 16 coll channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer
node:101:140 [0] NCCL INFO AllGather: 19009536 Bytes -> Algo 1 proto 2 time 512.1
node:101:140 [0] NCCL INFO AllGather: 19009536 Bytes -> Algo 1 proto 2 time 512.1
node:101:140 [0] NCCL INFO AllReduce: 40 Bytes -> Algo 0 proto 0 time 7.3
"""
NEW = """
h:7:9 [3] NCCL INFO NCCL version 2.21.5+hip6.3
h:7:9 [3] NCCL INFO Channel 03 : 3[3] -> 4[4] [send] via NET/Socket/0
h:7:9 [3] NCCL INFO Gather: 9504768 Bytes -> Algo RING proto LL128 channel{Lo..Hi}={0..7}
"""


def test_numeric_tuning_lines_transports_channels_and_forced_variables():
    r = bd.parse_nccl_debug(OLD)
    assert r["version"].startswith("2.18.6")
    assert r["collectives"]["AllGather"] == [{"bytes": 19009536, "algo": "RING", "proto": "SIMPLE"}]
    assert r["collectives"]["AllReduce"] == [{"bytes": 40, "algo": "TREE", "proto": "LL"}]
    assert r["transports"] == {"P2P/IPC/read": 3} and r["channels"] == 16
    assert r["forced"] == {"NCCL_ALGO": "Ring"}


def test_symbolic_tuning_lines():
    r = bd.parse_nccl_debug(NEW)
    assert r["version"].startswith("2.21.5")
    assert r["collectives"]["Gather"] == [{"bytes": 9504768, "algo": "RING", "proto": "LL128"}]
    assert r["transports"] == {"NET/Socket/0": 1}


def test_empty_log_and_environment():
    r = bd.parse_nccl_debug("")
    assert r["collectives"] == {} and r["version"] is None
    env = bd.nccl_debug_env("/tmp/x.log", algo="Ring")
    assert env["NCCL_DEBUG"] == "INFO" and env["NCCL_ALGO"] == "Ring" and env["NCCL_DEBUG_FILE"] == "/tmp/x.log" and "NCCL_PROTO" not in env


def test_version_line_of_the_rccl_of_this_image():
    """ROCm 7.0's RCCL (what `bench.py --gather-report` met on the MI355X box, one rank): `RCCL version : 2.26.6-HEAD:64f48b6`; a one-rank communicator logs no
    tuning line (nothing to choose)."""
    r = bd.parse_nccl_debug("runc:172:172 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6\nHIP version  : 7.0.51831-7c9236b16\n"
                            "runc:172:242 [0] NCCL INFO ncclCommInitRankConfig_impl comm 0x6151 rank 0 nranks 1 cudaDev 0 nvmlDev 0 busId d000 - Init START\n")
    assert r["version"] == "2.26.6-HEAD:64f48b6" and r["collectives"] == {}

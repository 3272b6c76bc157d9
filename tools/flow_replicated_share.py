"""Round 6: what fraction of a rank's device time in vpp_semi_dense_optical_flow_sharded is REPLICATED on every rank?
Starts benchmarks/flow_strip_bench with W ranks on ONE GPU (RCCL over the loopback socket transport, as tests/test_gpu_multi_rank.py does), rank 0 under
`rocprofv3 --kernel-trace`, and sums rank 0's kernel time by kind: the claim / descent kernels are sharded by flow-map row strips (a rank runs 1 / W of them), every
other kernel of the flow entry (pyramids, the propagation sweeps, the read-back) runs whole on every rank; RCCL's own kernels and the FAST-9 strip leg are listed
apart.  The harness runs with VPP_STRIP_SHARE_ONLY=1: only the sharded step (row exchange + vpp_semi_dense_optical_flow_sharded), 8 times.  usage: python tools/flow_replicated_share.py [W ...]    -> markdown on stdout (profiles/r06_flow_replicated_share.md)"""
import glob, os, sqlite3, subprocess, sys, tempfile
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "benchmarks", "flow_strip_bench")


def kind(name):
    n = name.replace("(anonymous namespace)::", "")
    if "sdof_descent" in n or "sdof_reset_claim" in n or "sdof_claim" in n: return "sharded (claims + descents)"
    if "sdof_" in n or "pyramid" in n or "pyr_down" in n: return "replicated (pyramids, sweeps, read-back)"
    if "fast9" in n: return "fast9 on strips (separate leg)"
    if "nccl" in n.lower() or "rccl" in n.lower(): return "rccl"
    return "other (copies, fills)"


print("# Round 6 — replicated share of the strip-sharded semi-dense flow (tools/flow_replicated_share.py: rank 0 of benchmarks/flow_strip_bench under rocprofv3 --kernel-trace, W ranks on one MI355X, 4K pair)\n")
for W in [int(a) for a in sys.argv[1:]] or [2, 8]:
    out = tempfile.mkdtemp(prefix=f"flowshare_{W}_", dir="/tmp")
    uid = os.path.join(out, "id")
    procs = []
    for r in range(W):
        env = dict(os.environ)
        env.update({"NCCL_HOSTID": f"vpp-one-gpu-rank-{r}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1", "NCCL_P2P_DISABLE": "1", "NCCL_SHM_DISABLE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "VPP_SHARD_EAGER": "1", "TMPDIR": "/tmp", "VPP_STRIP_SHARE_ONLY": "1"})
        cmd = [EXE, str(r), str(W), uid, "5", "2160", "3840"]
        if r == 0:
            cmd = ["rocprofv3", "--kernel-trace", "-d", os.path.join(out, "kt"), "-o", "r0", "--"] + cmd
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd="/tmp"))
    ok = True
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill(); o, e = p.communicate()
        ok = ok and p.returncode == 0
    dbs = glob.glob(os.path.join(out, "kt", "**", "*_results.db"), recursive=True)
    if not ok or not dbs:
        print(f"## {W} ranks: the run failed (rc {[p.returncode for p in procs]})\n"); continue
    con = sqlite3.connect(dbs[0])
    acc, names = defaultdict(float), defaultdict(lambda: defaultdict(float))
    for name, dur in con.execute("select name, duration from kernels"):
        acc[kind(name)] += dur; names[kind(name)][name.replace("(anonymous namespace)::", "").split("(")[0][:70]] += dur
    flow = acc["sharded (claims + descents)"] + acc["replicated (pyramids, sweeps, read-back)"]
    print(f"## {W} ranks\n\n| kind | rank 0 device time, ms (all steps) | share of the flow entry |\n|---|---|---|")
    for k in ("sharded (claims + descents)", "replicated (pyramids, sweeps, read-back)", "rccl", "fast9 on strips (separate leg)", "other (copies, fills)"):
        share = f"{acc[k] / flow:.1%}" if k.startswith(("sharded", "replicated")) and flow else ""
        print(f"| {k} | {acc[k] / 1e6:.3f} | {share} |")
    print()
    for k in ("sharded (claims + descents)", "replicated (pyramids, sweeps, read-back)"):
        print(f"{k}: " + ", ".join(f"{n} {d / 1e6:.3f}" for n, d in sorted(names[k].items(), key=lambda kv: -kv[1])[:6]))
    print()

#!/usr/bin/env python3
"""Two RCCL ranks of THIS library on one box (two processes, possibly one GPU): which environment lets
ncclCommInitRank succeed?  Prints one line per variant; NCCL_DEBUG=INFO logs go to gpurun_out/rccl_probe/."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from revrand_amd import parallel
comm = parallel.init_rccl_from_env(device=int(os.environ.get("LOCAL_RANK", "0")))
out = comm.allreduce_host(np.array([1.0 + comm.rank]))
comm.barrier()
print("OK rank %%d of %%d sum %%s" %% (comm.rank, comm.world, out.tolist()), flush=True)
comm.close()
''' % ROOT

VARIANTS = {
    "hostid": {"NCCL_HOSTID": "rank{r}"},
    "hostid_lo": {"NCCL_HOSTID": "rank{r}", "NCCL_SOCKET_IFNAME": "lo"},
    "hostid_lo_nop2p": {"NCCL_HOSTID": "rank{r}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_P2P_DISABLE": "1", "NCCL_SHM_DISABLE": "1"},
    "hostid_lo_noib": {"NCCL_HOSTID": "rank{r}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1", "NCCL_NET": "Socket"},
    "plain": {},
}


def main():
    outdir = os.path.join(ROOT, "gpurun_out", "rccl_probe")
    os.makedirs(outdir, exist_ok=True)
    which = sys.argv[1:] or list(VARIANTS)
    for name in which:
        tmp = tempfile.mkdtemp()
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", RR_COMM_RDZV="file:%s/id" % tmp,
                       NCCL_DEBUG="INFO", NCCL_DEBUG_FILE=os.path.join(outdir, "%s_rank%d.log" % (name, r)))
            for k, v in VARIANTS[name].items():
                env[k] = v.format(r=r)
            procs.append(subprocess.Popen([sys.executable, "-c", CODE], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True))
        res = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=120)
                res.append((p.returncode, o.strip().splitlines()[-1][-300:] if o.strip() else ""))
            except subprocess.TimeoutExpired:
                p.kill()
                res.append(("timeout", ""))
        print(json.dumps({"variant": name, "ranks": res}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Launch tools/two_rank_check.py as two ranks ON ONE GPU over the peer-to-peer transport (the parent never
touches the GPU).  usage: p2p_two_ranks.py [ENV=VALUE ...]   (e.g. AA_HIP_OPTIONS=proj_mode=1 TWO_RANK_ONLY_MAIN=1)"""
import os
import socket
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = dict(a.split("=", 1) for a in sys.argv[1:])
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
procs = []
for rank in range(2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AA_LAUNCH_ID=str(os.getpid()), AA_COMM="p2p",
               CONVEX_DIM_RED_DEVICE="0", **extra)
    procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tools", "two_rank_check.py")], env=env, cwd=root,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True))
rc = 0
for r, pr in enumerate(procs):
    try:
        out = pr.communicate(timeout=300)[0]
    except subprocess.TimeoutExpired:
        pr.kill()
        out = pr.communicate()[0] + "\nTIMEOUT"
    print("=== rank %d (exit %s) %s\n%s" % (r, pr.returncode, extra, out[-1800:]), flush=True)
    rc |= pr.returncode or 0
sys.exit(1 if rc else 0)

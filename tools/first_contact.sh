#!/bin/bash
# First contact with a multi-GPU node (run from the repo root; 20 minutes at most):
#     tools/first_contact.sh [tag]
#   1. the multi-device tests (tests/test_gpu_multi_device.py: real RCCL / peer mapping over xGMI; skipped on one device)
#   2. bench.py --gpus N --no-extra for N = 2, 4, 8 as far as the node has devices -- every rank checks its last step's
#      hit lists against a whole-range handle (on by default for N > 1), and the line says which transport ran
#      (config.transport, transport_ranks_seen, transport_note on any fall-back), what crossed a link
#      (exchange_gbs_per_link) and whether a batch had to be redone densely (exchange_redone_densely)
#   3. the four lines (N = 1 too) land in profiles/<tag>_first_contact_n<N>.json, the test log beside them
# On a ONE-GPU box the ranks share the device (ipc transport, gloo for bench.py's own barrier): N = 2 and 4 only -- the
# box allows six processes on its card.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
cd $R; mkdir -p gpurun_out profiles
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
left() { echo $(( 1200 - ($(date +%s) - T0) )); }
NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 1)
echo "[first contact] $NDEV device(s)"
timeout -k 10 $(( $(left) < 420 ? $(left) : 420 )) python3 -m pytest tests/test_gpu_multi_device.py -x -q -m gpu > gpurun_out/${TAG}_first_contact_tests.log 2>&1
echo "[first contact] multi-device tests: exit $? ($(tail -1 gpurun_out/${TAG}_first_contact_tests.log))"
cp gpurun_out/${TAG}_first_contact_tests.log profiles/${TAG}_first_contact_tests.log
if [ "$NDEV" -le 1 ]; then NS="1 2 4"; else NS="1"; for n in 2 4 8; do [ $n -le $NDEV ] && NS="$NS $n"; done; fi
for n in $NS; do
  [ $(left) -lt 90 ] && { echo "[first contact] out of time before N = $n"; break; }
  # (ranks that share ONE device also share its 288 GB: one resident query batch per rank instead of three)
  RING=""; [ "$NDEV" -le 1 ] && [ $n -ge 4 ] && RING="--ring 1"
  timeout -k 10 $(( $(left) < 400 ? $(left) : 400 )) python3 bench.py --gpus $n --steps 9 --warmup 2 --no-cpu --no-extra --no-pmc $RING \
      > gpurun_out/${TAG}_first_contact_n$n.json 2> gpurun_out/${TAG}_first_contact_n$n.err
  rc=$?
  if [ $rc -eq 0 ] && [ -s gpurun_out/${TAG}_first_contact_n$n.json ]; then
    cp gpurun_out/${TAG}_first_contact_n$n.json profiles/${TAG}_first_contact_n$n.json
    python3 - <<PY
import json
j = json.load(open("gpurun_out/${TAG}_first_contact_n$n.json"))
c = j["config"]
print("[first contact] N = %d: %.0f genomes/s, %.2f ms/step, transport %s (ranks seen %s%s), %s GB/s per link, redone densely %s, verify %s"
      % (j["n_gpus"], j["value"], j["ms_per_step"], c.get("transport"), c.get("transport_ranks_seen"),
         ", NOTE: " + c["transport_note"] if c.get("transport_note") else "", c.get("exchange_gbs_per_link"), c.get("exchange_redone_densely"),
         (j.get("verify") or {}).get("hit_lists_equal_whole_range_handle")))
PY
  else
    echo "[first contact] N = $n failed (exit $rc):"; tail -5 gpurun_out/${TAG}_first_contact_n$n.err
  fi
done
echo "[first contact] done in $(( $(date +%s) - T0 )) s"

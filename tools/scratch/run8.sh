for e in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=0" "AMD_SERIALIZE_KERNEL=3" "DEBUG_CLR_SKIP_RELEASE_SCOPE=0"; do
  echo "== $e"
  env $e python tools/scratch/dbg_loop8.py 2>&1 | grep "eager G_reg vs replay\|replay twice\|Error" | head -4
done

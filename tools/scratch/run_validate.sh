for e in "TAP=0" "TAP=1" "TAP=0 IGAN_GRAPH_VALIDATE_CONTEXT=0" "TAP=0 FMAP=1024" "TAP=0 FMAP=4096"; do
  echo "== $e"; env $e python tools/scratch/dbg_validate.py 2>&1 | grep "GRAPHS\|WARNING\|Error" | cut -c1-300
done

for h in '^$' '.' 'empty|zeros|ones|full' 'mul|add|sum|div|sub' 'view|reshape|expand|permute|slice|select|as_strided|t\.' 'copy|clone|contiguous|cat|stack|where|index'; do
  HOLD="$h" python tools/scratch/dbg_loop7.py 2>&1 | grep "^HOLD\|Error"
done
NAMES=1 HOLD='^$' python tools/scratch/dbg_loop7.py 2>&1 | grep "^\[" | cut -c1-3000

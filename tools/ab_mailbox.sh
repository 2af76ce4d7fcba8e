set -u
ROOT=$(pwd)
cd /tmp
for rep in 1 2; do
for lib in "" $ROOT/stwo-brainfuck_amd/libbfhip_before.so; do
echo "== lib=${lib:-new}"
BFHIP_LIBRARY=$lib python3 $ROOT/tools/merkle_shapes.py 2>&1 | grep -E "leaf 16|leaf 64|leaf 128|inner \+ 16|inner \+ 1 col|leaf 4 cols  "
for w in 22 fib19; do BFHIP_LIBRARY=$lib python3 $ROOT/tools/point.py $w --steps 30 | cut -c1-130; done
done; done

# scratch: tile configs on the T = 200 k5 convs of the DiT FFN (B = 32)
for c in M256 M128 M64 S64; do
  for shape in "192 768 5" "192 384 5" "192 768 3"; do
    set -- $shape
    echo -n "$c cin=$1 cout=$2 k=$3: "
    HSP_LIB=$PWD/variants/libhsp_$c.so python tools/conv_bench.py --cin $1 --cout $2 --k $3 --len 200 --batch 32 --act 0 --reps 50 2>&1 | tail -1
  done
done

for cfg in "3 5 4" "3 5 8" "3 5 16" "4 5 4" "4 5 8" "4 5 16" "2 5 8" "2 5 16" "5 5 8" "3 9 8" "3 3 8"; do
  echo -n "tail/run/bands $cfg: "; python tools/pyr_kt.py $cfg 2>/dev/null | grep pyr | awk '{s+=$2; printf "%s %.4f  ", $1, $2} END {printf " SUM %.4f\n", s}'
done

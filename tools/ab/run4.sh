for lv in 1 2 3 4; do echo "== LV2=$lv"; GSTARK_MERKLE_LV2=$lv python tools/merkle_ab.py 2>&1 | grep -E "x [46] cols" | grep -E "2\^(22|20|18)"; done

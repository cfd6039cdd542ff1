for b in base noprio k0; do echo "== $b"; timeout -k 10 60 build_micro/pt_dev_$b 30 32 50 1 0 | grep "new\|diff"; done
for t in 1 2 3 5 8; do echo "== tiles $t"; PT_TILES=$t timeout -k 10 60 build_micro/pt_dev_base 30 32 50 1 0 | grep "ptd\|diff"; done

# usage (GPU box): scripts/ab_enc.sh <lib name|default> ...  -- kbench encoder case per library build
for lib in "$@"; do
  if [ "$lib" = default ]; then unset ZIRA_MSDA_LIB; else export ZIRA_MSDA_LIB=$PWD/build_ab/$lib.so; fi
  echo "== $lib: $(CASES=encoder ROUNDS=3 timeout 300 python scripts/kbench.py 2>&1 | grep encoder)"
done

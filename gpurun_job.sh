python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x -k "batch or golden" 2>&1 | tail -2
python tools/bench_configs.py 2>&1 | grep -E "cfg5|Error|error" 

python -m pytest tests/test_gpu_predict.py -m gpu -q --no-header -p no:cacheprovider -k "predict_vs_oracle" 2>&1 | tail -40

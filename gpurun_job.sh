python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -25

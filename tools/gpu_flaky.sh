#!/bin/bash
# repeat the -m gpu suite to catch statistically flaky tests
for i in $(seq 1 ${1:-12}); do
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 | grep -E "failed|passed|FAILED" | tr '\n' ' '; echo
done

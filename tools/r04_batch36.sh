#!/bin/bash
CCP_GS_DEBUG=1 timeout -k 10 200 python -m pytest tests/test_gpu_lex.py -m gpu -x -q -s -k "heads" 2>&1 | grep -v "^  File\|Extension\|^$" | tail -25

#!/bin/bash
timeout 300 python scripts/enc_grad_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-250

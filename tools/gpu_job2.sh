#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout -k 10 300 python tools/debug_rng.py 2>&1 | grep -v amdgpu.ids | tail -10

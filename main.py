#!/usr/bin/env python3
"""Entry point mirroring the reference's `python main.py <command>` (/root/reference/main.py:25-31)."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # see unet-watermark_amd/_lib.py

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from unet_watermark_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()

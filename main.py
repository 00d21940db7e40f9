#!/usr/bin/env python3
"""Entry point mirroring the reference's `python main.py <command>` (/root/reference/main.py:25-31)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from unet_watermark_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Drop-in for the reference's `python main.py ...` (see ultrare_amd/main.py)."""
from ultrare_amd.main import main

if __name__ == '__main__':
    main()

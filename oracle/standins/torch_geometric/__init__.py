"""TEST-ONLY stand-in for torch_geometric 2.0.2. Not product code."""

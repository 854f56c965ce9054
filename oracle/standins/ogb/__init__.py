"""TEST-ONLY stand-in for ogb==1.2.1 encoders. Not product code."""

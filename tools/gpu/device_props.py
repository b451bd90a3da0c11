#!/usr/bin/env python3
"""What the device says about itself: the numbers fpv_create checks the cache model against (hipGetDeviceProperties)."""
import json
import torch
p = torch.cuda.get_device_properties(0)
d = {k: getattr(p, k) for k in dir(p) if not k.startswith("_") and isinstance(getattr(p, k), (int, float, str, bool))}
print(json.dumps(d, indent=1, default=str))

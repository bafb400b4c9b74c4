import torch
try:
    print("range", torch.cuda.Stream.priority_range())
except Exception as e:
    print("no priority_range", e)
for p in (-2,-1,0,1,2,3):
    try:
        s = torch.cuda.Stream(priority=p); print(p, "ok", s.priority)
    except Exception as e:
        print(p, "err", str(e)[:80])

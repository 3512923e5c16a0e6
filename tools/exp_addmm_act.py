import torch, time
dev = torch.device("cuda:0")
for rows, cin, cout in ((1048576, 64, 64), (1048576, 64, 128), (524288, 131, 128), (524288, 128, 256)):
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cout, cin, device=dev); b = torch.randn(cout, device=dev)
    def f1(): return torch.relu(torch.nn.functional.linear(x, w, b))
    def f2(): return torch._addmm_activation(b, x, w.t(), use_gelu=False)
    def f3(): return torch.nn.functional.linear(x, w, b).relu_()
    for nm, f in (("linear+relu", f1), ("addmm_activation", f2), ("linear+relu_", f3)):
        try:
            y = f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            err = (y - f1()).abs().max().item()
            print(rows, cin, cout, nm, round(e0.elapsed_time(e1) * 100, 1), "us", "maxdiff", err)
        except Exception as ex:
            print(nm, "failed:", repr(ex)[:200])

"""P ranks of ONE process for the GPU tests of the C ABI's optimiser loops on sharded handles and split pieces."""


class ThreadRanks:
    """P ranks of one process, one Python thread each, for the optimiser loops of the C ABI on sharded handles
    (gpslam_hip_set_collectives): the host's all_gather / all_reduce_sum are played by a barrier and device copies.  ctypes releases
    the GIL for the duration of a library call and takes it back for the callbacks, so the P calls really run side by side and
    meet in the collectives, as P processes would."""

    def __init__(self, P):
        import threading
        self.P = P
        self.bar = threading.Barrier(P)
        self.slots = [None] * P

    def collectives(self, r):
        import torch
        from gpslam_amd.sharded import _DevView
        P, bar, slots = self.P, self.bar, self.slots

        def view(ptr, nbytes):
            return torch.as_tensor(_DevView(ptr, nbytes), device="cuda")

        def all_gather(send, recv, nbytes, _stream):
            torch.cuda.synchronize()                 # this rank's record is complete (every handle has its own stream)
            slots[r] = (send, nbytes)
            bar.wait()
            out = view(recv, nbytes * P).view(P, -1)
            for k in range(P):
                out[k].copy_(view(*slots[k]))
            torch.cuda.synchronize()
            bar.wait()                               # nobody rewrites its record before everybody has read it

        def all_reduce_sum(buf, n, _stream):
            torch.cuda.synchronize()
            slots[r] = (buf, n * 8)
            bar.wait()
            total = sum(view(*slots[k]).clone() for k in range(P))      # rank order on every rank: identical sums
            torch.cuda.synchronize()
            bar.wait()
            view(buf, n * 8).copy_(total)
            torch.cuda.synchronize()
            bar.wait()

        return all_gather, all_reduce_sum

    def run(self, fn):
        """fn(rank) on P threads; returns the list of results, re-raises the first exception"""
        import threading
        out, err = [None] * self.P, []

        def body(r):
            try:
                out[r] = fn(r)
            except BaseException as e:      # noqa: BLE001
                err.append(e)
                self.bar.abort()
        ts = [threading.Thread(target=body, args=(r,)) for r in range(self.P)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        if err:
            raise err[0]
        return out

"""Collecting results from worker processes of a multi-process test without waiting out the full timeout when a
worker has died: a rank that raises leaves its peers blocked in a collective, and the parent then sat silent for
600-900 s (long enough for a GPU box's silence watchdog to kill the whole run) before the test failed."""
import queue
import time


def gather_results(q, procs, count, timeout):
    """`count` items from queue `q`, or an AssertionError naming the workers that exited with an error / the timeout;
    the surviving workers are terminated (exactly the processes in `procs`)."""
    out = []
    deadline = time.time() + timeout
    while len(out) < count:
        try:
            out.append(q.get(timeout=1.0))
            continue
        except queue.Empty:
            pass
        dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
        late = time.time() > deadline
        if dead or late:
            for p in procs:
                if p.is_alive():
                    p.terminate()
            for p in procs:
                p.join(timeout=10)
            raise AssertionError(f"worker(s) exited with an error: {dead}" if dead else f"no result from every worker within {timeout} s")
    return out


# ---- rendezvous of the workers through a FILE store -----------------------------------------------------------------
# The tests used to pick a free TCP port in the parent (bind to port 0, close) and hand it to the workers as MASTER_PORT:
# between the close and rank 0's listen another socket of a long test session can take the port, rank 0 then raises,
# its peers wait in the rendezvous, the parent in q.get.  A file store has no such window.  `token` is any number that is
# unique among the concurrently running tests of one session (the tests keep passing the port number they used to pick).
def store_path(token, parent_pid):
    return f"/tmp/zk_test_store_{parent_pid}_{token}"


def fresh_store(token):
    """Parent, before the workers start: no stale store of an earlier session with the same pid and token."""
    import os
    try:
        os.unlink(store_path(token, os.getpid()))
    except OSError:
        pass


def init_pg(backend, token, rank, world, **kw):
    """Worker: torch.distributed.init_process_group over the file store of this test (the parent is the spawning process)."""
    import datetime
    import os
    import torch.distributed as dist
    dist.init_process_group(backend, init_method=f"file://{store_path(token, os.getppid())}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=300), **kw)

"""``pool=`` object for dynesty that turns a queue fill into one device launch.

dynesty needs only ``.map(func, iterable)`` and ``.size`` from a pool
(reference utils.py:2358-2381, pool.py:148-173).  ``Sampler._fill_queue`` calls
``pool.map(internal_sampler.sample, args)`` with ``queue_size`` arguments that
share one ``loglstar`` (sampler.py:701-717): when ``func`` is one of our
samplers' ``sample`` the whole list is executed by a single kernel, and so is
dynesty's own ``UnitCubeSampler.sample`` (the phase before the first bound) when
the run's callbacks are a device Problem's; anything else (prior transforms and
likelihood calls of the initial live points, bootstrap replicas) is mapped
serially on the host like ``map`` would.
"""


def _is_dynesty_unitcube(func):
    """True only for dynesty's own ``internal_samplers.UnitCubeSampler.sample`` -- by identity against the loaded
    module (no import here: if ``func`` is dynesty's, dynesty is in ``sys.modules``).  A user class that merely
    shares the qualified name keeps its own semantics and is mapped serially."""
    if getattr(func, '__qualname__', '') != 'UnitCubeSampler.sample':
        return False
    import sys
    mod = sys.modules.get('dynesty.internal_samplers')
    cls = getattr(mod, 'UnitCubeSampler', None)
    return cls is not None and getattr(func, '__func__', func) is getattr(cls.sample, '__func__', cls.sample)


class HipBatchPool:

    def __init__(self, queue_size=1024):
        self.size = int(queue_size)

    def map(self, func, iterable):
        runner = getattr(func, '_dynhip_batch', None)
        if runner is not None:
            return runner(list(iterable))
        if _is_dynesty_unitcube(func):
            # dynesty's own sampler of the phase before the first bound (it constructs it itself, so there is no
            # drop-in class to hand it): batched when the run's callbacks are a device Problem's.  The batched cube
            # phase evaluates prior transform and ln L with the DEVICE twin's arithmetic (ocml), the serial form below
            # with the host Problem's (NumPy): equal to ~1e-15 relative, not bit for bit (tests hold 1e-10).
            from . import samplers
            args = list(iterable)
            res = samplers.run_unitcube(args)
            return res if res is not None else list(map(func, args))
        return list(map(func, iterable))

    def close(self):
        pass

    def join(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

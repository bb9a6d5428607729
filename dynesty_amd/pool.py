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


class HipBatchPool:

    def __init__(self, queue_size=1024):
        self.size = int(queue_size)

    def map(self, func, iterable):
        runner = getattr(func, '_dynhip_batch', None)
        if runner is not None:
            return runner(list(iterable))
        if getattr(func, '__qualname__', '') == 'UnitCubeSampler.sample':
            # dynesty's own sampler of the phase before the first bound (it constructs it itself, so there is no
            # drop-in class to hand it): batched when the run's callbacks are a device Problem's
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

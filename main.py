"""Drop-in for the reference's ``main.py`` (``main.py:8-187``): same flags, result directories, file
formats and metric line, on the MI355X path.  Implementation: ``dorpatch_amd/driver.py``.

    python main.py --targeted --patch_budget 0.12                      # 1 GPU, like the reference
    python main.py --synthetic --num_images 2 --max_iterations 50      # offline: no ImageNet / checkpoint
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        main.py --targeted --num_images 1000 --max_iterations 1000 --shard images      # BASELINE configs[4]
"""
from dorpatch_amd.driver import build_parser, main, run  # noqa: F401

parser = build_parser()          # the reference exposes a module-level ``parser`` (main.py:8)

if __name__ == '__main__':
    main()

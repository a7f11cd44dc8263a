"""CPU restatement of the reference's RRDBNet/ESRGAN path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker.  The product (image_restoration_amd) never does.
"""

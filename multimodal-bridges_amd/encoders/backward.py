"""placeholder -- replaced below"""
def attach_backward(model, loss, tape, heads, d_hc, d_hd, d_s):
    return loss[0]

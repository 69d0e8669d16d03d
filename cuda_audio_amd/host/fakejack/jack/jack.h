/* jack/jack.h — offline stand-in for the subset of the JACK C API the host uses
 * (reference src/jackclient.h:4-5 includes <jack/jack.h>; libjack is not in
 * this image).  Same names, types and call order as libjack for:
 * client open/close, process + shutdown callbacks, port register / buffer /
 * name, activate, connect, sample rate.  A driver entry (fakejack_run) plays
 * the role of jackd: it fills the input port buffers and invokes the process
 * callback once per period on the calling thread. */
#ifndef FAKE_JACK_H
#define FAKE_JACK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint32_t jack_nframes_t;
typedef float jack_default_audio_sample_t;
typedef struct fake_jack_client jack_client_t;
typedef struct fake_jack_port jack_port_t;
typedef int jack_options_t;
typedef int jack_status_t;
typedef int (*JackProcessCallback)(jack_nframes_t nframes, void* arg);
typedef void (*JackShutdownCallback)(void* arg);

enum { JackNullOption = 0, JackNoStartServer = 1 };
enum { JackFailure = 1, JackNameNotUnique = 4, JackServerFailed = 0x10 };
enum { JackPortIsInput = 1, JackPortIsOutput = 2 };
#define JACK_DEFAULT_AUDIO_TYPE "32 bit float mono audio"
#define JACK_DEFAULT_MIDI_TYPE "8 bit raw midi"

jack_client_t* jack_client_open(const char* name, jack_options_t options, jack_status_t* status, ...);
int jack_client_close(jack_client_t* c);
int jack_set_process_callback(jack_client_t* c, JackProcessCallback cb, void* arg);
void jack_on_shutdown(jack_client_t* c, JackShutdownCallback cb, void* arg);
jack_nframes_t jack_get_sample_rate(jack_client_t* c);
jack_nframes_t jack_get_buffer_size(jack_client_t* c);
jack_port_t* jack_port_register(jack_client_t* c, const char* port_name, const char* port_type, unsigned long flags,
                                unsigned long buffer_size);
void* jack_port_get_buffer(jack_port_t* p, jack_nframes_t nframes);
const char* jack_port_name(const jack_port_t* p);
int jack_activate(jack_client_t* c);
int jack_connect(jack_client_t* c, const char* source_port, const char* destination_port);

/* ---- driver side (what jackd would do) -------------------------------- */
/* sample rate / period used by clients opened afterwards (defaults 44100 / 256) */
void fakejack_configure(jack_nframes_t sample_rate, jack_nframes_t period);
/* Run `nperiods` periods: for each, `feed(period_index, inputs, n_in, nframes, user)`
 * fills the client's input port buffers (registration order), the process
 * callback runs, then `drain(period_index, outputs, n_out, nframes, user)` sees
 * the output buffers.  Returns the number of periods processed. */
typedef void (*fakejack_io_fn)(uint64_t period, float** bufs, size_t nbufs, jack_nframes_t nframes, void* user);
uint64_t fakejack_run(jack_client_t* c, uint64_t nperiods, fakejack_io_fn feed, fakejack_io_fn drain, void* user);
/* The same for several clients AT ONCE, as jackd runs them: every client gets a driver thread of its own that calls its process
 * callback once per period (jackclient.cu:4-11: one real-time thread per client; main.cu:31-39 opens conv.count / 2 of
 * them on one GPU).  users[i] is passed to client i's feed / drain.  spacing_us > 0: a thread starts period k no earlier than
 * k * spacing_us after the common start (the period clock; 0 = back to back).  us_per_period[i] (may be null) receives client i's
 * mean time inside its process callback.  Returns the periods every client completed. */
uint64_t fakejack_run_all(jack_client_t** clients, size_t nclients, uint64_t nperiods, fakejack_io_fn feed, fakejack_io_fn drain,
                          void** users, double spacing_us, double* us_per_period);
/* simulate the server going away: invokes the shutdown callback */
void fakejack_shutdown(jack_client_t* c);

#ifdef __cplusplus
}
#endif
#endif

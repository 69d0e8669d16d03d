/* jack/midiport.h — empty stand-in (the reference includes it, src/jackclient.h:5, but uses nothing from it) */
#ifndef FAKE_JACK_MIDIPORT_H
#define FAKE_JACK_MIDIPORT_H
#endif
